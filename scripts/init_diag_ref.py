"""Development aid (build container only: imports /root/reference): what the REFERENCE's two-frame initialisation sees on the
first frames of the rendered 640x480 sequence of tests/golden/ate_sequence_640.npz -- per frame: overlap fraction, |t|, median
depth, the decision.  The same numbers of the HIP loop: scripts/init_diag.py (GPU box)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))
import make_golden_r2 as r2  # noqa: E402  (sets up sys.path, shims)
from como.odom.frontend import TwoFrameSfm as tfm  # noqa: E402
from como.utils.coords import fill_image  # noqa: E402

nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
orig = tfm.TwoFrameSfm.handle_frame


def handle_frame(self, rgb, timestamp):
    had = self.has_reference
    out = orig(self, rgb, timestamp)
    if had:
        T, depth = out[1], out[5]
        reproj = fill_image(out[4], depth, self.img_and_grads[-1].shape[-2:])
        seen = torch.count_nonzero(~torch.isnan(reproj)).item()
        print(f"DIAG ts {timestamp}: is_init {out[0]} has_ref_after {self.has_reference} seen/px {seen}/{self.vals_pyr[-1].shape[2]} "
              f"|t| {torch.linalg.norm(T[:, :3, 3]).item():.6f} med_depth {torch.median(depth).item():.6f} "
              f"mean_logd {float(out[6]):.6f} t {T[0, :3, 3].tolist()}", flush=True)
    else:
        print(f"DIAG ts {timestamp}: new reference; coords_m[:3] {self.coords_m[0, :3].tolist()}", flush=True)
    return out


tfm.TwoFrameSfm.handle_frame = handle_frame
torch.set_num_threads(8)
step = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
deg = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fs = float(sys.argv[5]) if len(sys.argv) > 5 else None
r2.ate_case(seed=seed, H=480, W=640, nframes=nframes, step=step, deg=deg, network_size=(192, 256), freq_scale=fs)
