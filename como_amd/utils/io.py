"""Trajectory output in the TUM format (reference como/utils/io.py:4-23, geometry/lie_algebra.py:11-42): one line per pose,
`timestamp tx ty tz qx qy qz qw`, four decimals.  Quaternions through scipy's Rotation, as the reference."""
import numpy as np
from scipy.spatial.transform import Rotation


def pose_to_tq(pose):
    """(4,4) or (B,4,4) array-like -> (7,) or (B,7): translation, then the unit quaternion (x, y, z, w)."""
    pose = np.asarray(pose.detach().cpu() if hasattr(pose, "detach") else pose, dtype=np.float64)
    q = Rotation.from_matrix(pose[..., :3, :3]).as_quat()
    return np.concatenate([pose[..., :3, 3], q], axis=-1)


def tq_to_pose(tq):
    tq = np.asarray(tq, dtype=np.float64)
    T = np.zeros(tq.shape[:-1] + (4, 4))
    T[..., :3, :3] = Rotation.from_quat(tq[..., 3:]).as_matrix()
    T[..., :3, 3] = tq[..., :3]
    T[..., 3, 3] = 1.0
    return T


def save_traj(filename, timestamps, poses):
    tq = pose_to_tq(poses)
    with open(filename, "w") as f:
        for ts, row in zip(timestamps, tq.reshape(-1, 7)):
            f.write("%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n" % ((float(ts),) + tuple(row)))
