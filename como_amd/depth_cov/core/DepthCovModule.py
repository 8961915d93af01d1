"""DepthCovModule mirror (como/depth_cov/core/DepthCovModule.py:15-87) + Mapping.run_model (Mapping.py:397-428).

`DepthCovModule(state_dict)` takes the reference checkpoint's ``state_dict`` (keys ``gaussian_cov_net.*``,
``cov_modules.N.scale_param``, ``log_depth_var_scales.N``); missing scale / variance parameters default to 0 as in the
reference constructor.
"""
import math

import torch

from como_amd import _lib
from como_amd.depth_cov.core import covariance as cv
from como_amd.depth_cov.nn import UNet as unet


class DepthCovModule:
    num_levels = 5
    depth_var_prior = 1e-2
    kernel_scale_prior = 1e0

    def __init__(self, state_dict):
        self.net = unet.UNet(state_dict, num_levels=self.num_levels, prefix="gaussian_cov_net.")
        n = self.num_levels - 1
        self.scale_params = [float(state_dict.get(f"cov_modules.{i}.scale_param", 0.0)) for i in range(n)]
        self.log_depth_var_scales = [float(state_dict.get(f"log_depth_var_scales.{i}", 0.0)) for i in range(n)]
        # the per-level module lists Mapping.prep_predictor / the sampler index (`model.cov_modules[level](...)`)
        self.cov_modules = [cv.CovarianceModule(self.get_scale(i)) for i in range(n)]
        self.cross_cov_modules = [cv.CrossCovarianceModule(self.get_scale(i)) for i in range(n)]
        self.diagonal_cov_modules = [cv.DiagonalCovarianceModule(self.get_scale(i)) for i in range(n)]

    def get_var(self, level):
        return self.depth_var_prior * math.exp(self.log_depth_var_scales[level])

    def get_scale(self, level):
        return self.kernel_scale_prior * math.exp(self.scale_params[level])

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def forward(self, rgb, finest_only=False):
        """(N,3,H,W) float in [0,1] -> list of 4 covariance images (N,4,h,w), coarse to fine (DepthCovModule.py:80-87).
        finest_only: only the last entry is evaluated (the others are None): `Mapping.run_model` reads `[-1]` alone, the three
        coarser heads (a 1x1 convolution + the activation each) are dead work on that path."""
        return [None if f is None else unet.cov_activation(f) for f in self.net(rgb, finest_only=finest_only)]

    __call__ = forward

    def forward_graphed(self, rgb, finest_only=False):
        """Same result through a hipGraph captured once per input shape (~95 launches replayed without the host in the
        loop).  The returned tensors are the graph's static outputs: consume them before the next call."""
        key = (tuple(rgb.shape), str(rgb.device), bool(finest_only))
        ent = self._graphs.get(key) if hasattr(self, "_graphs") else None
        if ent is None:
            if not hasattr(self, "_graphs"):
                self._graphs = {}
            x = torch.empty(rgb.shape, dtype=torch.float32, device=rgb.device)
            x.copy_(rgb)
            side = torch.cuda.Stream(device=rgb.device)
            side.wait_stream(torch.cuda.current_stream(rgb.device))
            with torch.cuda.stream(side):
                self.forward(x, finest_only)
            torch.cuda.current_stream(rgb.device).wait_stream(side)
            torch.cuda.synchronize(rgb.device)
            g, outs = _lib.capture_graph(lambda: self.forward(x, finest_only), rgb.device)     # (on failure the eager path keeps working)
            ent = (g, x, outs if g is not None else None)
            self._graphs[key] = ent
        g, x, outs = ent
        if g is None:
            return self.forward(rgb, finest_only)
        x.copy_(rgb)
        g.replay()
        return outs


OUTPUT_HOOK = None      # tests only: cov -> cov applied to the network's output (the robustness test perturbs it by 1e-6 relative)


def run_model(model, rgb, network_size=(192, 256), dtype=torch.float64, graphed=True):
    """Mapping.run_model (Mapping.py:409-428): antialiased resize to the network size, finest covariance level, cast to
    the mapping dtype, antialiased resize back to the image size."""
    rgb_r = unet.resize_aa(rgb.float(), network_size)
    cov = (model.forward_graphed(rgb_r, finest_only=True) if graphed else model(rgb_r, finest_only=True))[-1].to(dtype)
    if OUTPUT_HOOK is not None:
        cov = OUTPUT_HOOK(cov)
    return unet.resize_aa(cov, rgb.shape[-2:])
