"""Image gradients and pyramids on the HIP kernels of csrc/image.hip.

Mirror of como/utils/image_processing.py (ImageGradientModule :8-44, GaussianBlurModule :47-65, ImagePyramidModule :68-87,
IntrinsicsPyramidModule :109-123): same class names, constructor arguments and call results.  GPU tensors only.
"""
import torch

from como_amd import _lib
from como_amd.geometry.camera import resize_intrinsics


def img_and_grads(img):
    """(N,C,H,W) -> (N,3C,H,W) = cat(img, gx, gy): the tensor Mapping.get_img_and_grads (Mapping.py:369-379) assembles."""
    _lib.require_cuda(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    out = torch.empty((N, 3 * C, H, W), dtype=img.dtype, device=img.device)
    fn = getattr(_lib.lib(), "como_img_grads_" + _lib.suffix(img.dtype))
    _lib.check(fn(img.data_ptr(), out.data_ptr(), N, C, H, W, _lib.stream_ptr(img.device)), "como_img_grads")
    return out


def blur_down(img):
    """GaussianBlurModule followed by [:, :, 0::2, 0::2] (one pyramid step, image_processing.py:85)."""
    _lib.require_cuda(img)
    img = img.contiguous()
    N, C, H, W = img.shape
    out = torch.empty((N, C, (H + 1) // 2, (W + 1) // 2), dtype=img.dtype, device=img.device)
    fn = getattr(_lib.lib(), "como_img_blur_down_" + _lib.suffix(img.dtype))
    _lib.check(fn(img.data_ptr(), out.data_ptr(), N * C, H, W, _lib.stream_ptr(img.device)), "como_img_blur_down")
    return out


class ImageGradientModule:
    def __init__(self, channels, device, dtype):
        self.channels = channels

    def __call__(self, x):
        C = x.shape[1]
        s = img_and_grads(x)
        return s[:, C:2 * C], s[:, 2 * C:]

    forward = __call__


class ImagePyramidModule:
    def __init__(self, channels, start_level, end_level, device, dtype):
        self.start_level, self.end_level = start_level, end_level

    def __call__(self, x):
        pyr = []
        lvl = x
        for i in range(self.end_level - 1):
            if i >= self.start_level:
                pyr.insert(0, lvl)
            lvl = blur_down(lvl)
        pyr.insert(0, lvl)
        return pyr

    forward = __call__


class IntrinsicsPyramidModule:
    def __init__(self, start_level, end_level, device):
        self.start_level, self.end_level = start_level, end_level

    def __call__(self, K_orig, image_scale_start):
        pyr = []
        for i in range(self.start_level, self.end_level):
            pyr.insert(0, resize_intrinsics(K_orig, [image_scale_start[0] * pow(2.0, -i), image_scale_start[1] * pow(2.0, -i)]))
        return pyr

    forward = __call__
