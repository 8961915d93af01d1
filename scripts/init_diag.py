"""At which frame does the two-frame initialisation of the rendered 640x480 bench sequence succeed?  (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
from como_amd import synth
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
from como_amd.odom.sequential import ComoSeq
from scripts.gpu_odometry_bench import cfgs
dev = "cuda:0"
H, W, frames = 480, 640, 14
scene = synth.PlaneScene(seed=1, freq_scale=1.0)
K = synth.intrinsics_for(H, W)
T = synth.gt_poses(frames, step=0.01, deg=0.3)
g = torch.Generator().manual_seed(1)
rgbs = []
for k in range(frames):
    I, _ = scene.render(T[k], K, H, W)
    I = I + 0.002 * torch.randn(I.shape, generator=g, dtype=torch.float64)
    rgbs.append(I[None, None].repeat(1, 3, 1, 1).to(dev))
model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
odo = ComoSeq(cfgs(dev, argparse.Namespace(pix="float")), K.clone(), (H, W), model)
first = None
for k in range(frames):
    odo.iter(1.0 + 0.033 * k, rgbs[k])
    if first is None and odo.mapping.is_init:
        first = k
print("init at frame", first, {k: os.environ.get(k) for k in ("COMO_SE3_KERNEL", "COMO_GRAM_KERNEL", "COMO_PIX_MIRRORS", "COMO_CHOL_SMALL_FAST")}, flush=True)
