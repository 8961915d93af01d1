"""Development aid (GPU box): what the HIP loop's two-frame initialisation sees on the first frames of the rendered 640x480
sequence of tests/golden/ate_sequence_640.npz -- per frame: overlap fraction, |t|, median depth, the decision.  The same numbers
of the reference's loop: scripts/init_diag_ref.py (build container).   python scripts/init_diag.py [frames] [step] [deg] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from como_amd.odom.frontend import TwoFrameSfm as tfm
from como_amd.utils.coords import fill_image
from scripts.ate_sequence import run_ate_sequence

nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 9
step = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
deg = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fs = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
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
G = {"H": 480, "W": 640, "nframes": nframes, "seed": seed, "step": step, "deg": deg, "colour": 0, "network_size": [192, 256], "freq_scale": fs}
kinds, poses, odo = run_ate_sequence(G, os.environ.get("PIX", "double"))
print("kinds", kinds)
