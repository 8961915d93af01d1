"""rocprofv3 helper: 5 eager forwards of the DepthCov network at 192x256 (development aid)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from como_amd import synth
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
DEV = "cuda:0"
model = DepthCovModule(synth.depthcov_state_dict(0, device=DEV))
x = torch.rand(1, 3, 192, 256, device=DEV)
for _ in range(5):
    model(x)
torch.cuda.synchronize()
