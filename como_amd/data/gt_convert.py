"""Ground-truth trajectories -> TUM format (reference scripts/convert_replica_gt.py:1-29, scripts/convert_scannet_test_gt.py:1-37).

    python -m como_amd.data.gt_convert replica <traj_dir/>      # reads traj.txt (one flattened 4x4 T_wc per line)
    python -m como_amd.data.gt_convert scannet <traj_dir/>      # reads pose/<i>.txt (4x4 T_wc; frames with non-finite poses dropped)
Both write `<traj_dir>traj_tum.txt` with timestamps i / 30 s (`como_amd.utils.io.save_traj`)."""
import glob
import os
import re
import sys

import numpy as np

from como_amd.utils.io import save_traj


def convert_replica_traj(traj_dir):
    """convert_replica_gt.py:7-19: every row of traj.txt is a row-major 4x4 camera-to-world pose; 30 Hz timestamps."""
    T_flat = np.loadtxt(os.path.join(traj_dir, "traj.txt")).reshape(-1, 16)
    T_wc = T_flat.reshape(-1, 4, 4)
    ts = (1.0 / 30.0) * np.arange(T_wc.shape[0])
    out = traj_dir + "traj_tum.txt"
    save_traj(out, ts, T_wc)
    return out


def convert_scannet_traj(traj_dir):
    """convert_scannet_test_gt.py:11-26: pose/<frame>.txt in numeric order; frames whose pose has a non-finite entry are
    skipped but keep their place on the 30 Hz time axis."""
    files = glob.glob(os.path.join(traj_dir, "pose/*.txt"))
    files = sorted(files, key=lambda x: int(re.findall(r"\d+", x.rsplit("/", 1)[-1])[0]))
    poses, ts = [], []
    for i, f in enumerate(files):
        p = np.loadtxt(f)
        if np.isfinite(p).all():
            poses.append(p)
            ts.append((1.0 / 30.0) * i)
    out = traj_dir + "traj_tum.txt"
    save_traj(out, ts, np.asarray(poses).reshape(-1, 4, 4))
    return out


if __name__ == "__main__":
    kind, d = sys.argv[1], sys.argv[2]
    print({"replica": convert_replica_traj, "scannet": convert_scannet_traj}[kind](d))
