"""Every convolution shape of the DepthCov network at 192x256 against torch's conv2d on the same GPU (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from como_amd.depth_cov.nn import UNet as U
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
shapes = [(3, 16, 3, 0), (16, 16, 3, 0), (3, 16, 1, 0)]
for i in range(5):
    c = 16 << i
    shapes += [(c, 2 * c, 3, i + 1), (2 * c, 2 * c, 3, i + 1), (c, 2 * c, 1, i + 1)]          # down block at level i+1
    shapes += [(2 * c, c, 3, i), (2 * c, c, 3, i), (c, c, 3, i), (2 * c, c, 1, i), (c, 3, 1, i)]   # up conv, up block, feature at level i
worst = 0.0
for (cin, cout, k, lvl) in shapes:
    H, W = 192 >> lvl, 256 >> lvl
    x = torch.randn(1, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)
    errs = []
    for rep in range(3):
        got = U._Conv(w, b)(x)
        errs.append(((got.double() - ref).abs().max() / ref.abs().max()).item())
    flag = "  <-- MISMATCH" if max(errs) > 1e-4 else ""
    worst = max(worst, max(errs))
    print(f"cin {cin:4d} cout {cout:4d} k {k} {H:3d}x{W:3d}: rel err {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}{flag}", flush=True)
print("worst", worst)
