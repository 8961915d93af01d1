"""Absolute trajectory error between two pose sequences (the "ATE vs ref" half of BASELINE.json's metric).

The reference ships no evaluation code (its README points to evo on the TUM-format files `como/utils/io.py` writes); this is
the standard definition evo implements: associate by timestamp, optionally align with the closed-form Umeyama transform
(SE(3), or Sim(3) for monocular scale), RMSE of the translation differences."""
import numpy as np


def umeyama(src, dst, with_scale):
    """Least-squares similarity transform dst ~ s R src + t for (N,3) point sets (Umeyama 1991)."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / src.shape[0]
    U, S, Vt = np.linalg.svd(cov)
    d = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        d[2] = -1.0
    R = U @ np.diag(d) @ Vt
    var = (xs ** 2).sum() / src.shape[0]
    s = float((S * d).sum() / var) if with_scale and var > 0 else 1.0
    t = mu_d - s * R @ mu_s
    return s, R, t


def ate_rmse(est_poses, ref_poses, align="none"):
    """est_poses, ref_poses: (N,4,4) array-likes of world poses at the SAME timestamps.
    align: "none" (both trajectories share their start frame and scale), "se3" or "sim3".
    Returns the RMSE of the translation differences (metres of the reference trajectory)."""
    est = np.asarray([np.asarray(p.detach().cpu() if hasattr(p, "detach") else p, dtype=np.float64).reshape(4, 4) for p in est_poses])
    ref = np.asarray([np.asarray(p.detach().cpu() if hasattr(p, "detach") else p, dtype=np.float64).reshape(4, 4) for p in ref_poses])
    if est.shape != ref.shape or est.shape[0] == 0:
        raise ValueError("ate_rmse: need two non-empty pose sequences of equal length")
    pe, pr = est[:, :3, 3], ref[:, :3, 3]
    if align != "none":
        s, R, t = umeyama(pe, pr, with_scale=(align == "sim3"))
        pe = (s * (R @ pe.T)).T + t
    return float(np.sqrt(((pe - pr) ** 2).sum(1).mean()))


def associate(ts_a, ts_b, max_dt=0.02):
    """Greedy nearest-timestamp association (TUM tools semantics): list of (i, j) with |ts_a[i] - ts_b[j]| < max_dt, each used once."""
    ts_a, ts_b = np.asarray(ts_a, dtype=np.float64), np.asarray(ts_b, dtype=np.float64)
    cand = sorted((abs(a - b), i, j) for i, a in enumerate(ts_a) for j, b in enumerate(ts_b) if abs(a - b) < max_dt)
    used_a, used_b, out = set(), set(), []
    for _, i, j in cand:
        if i not in used_a and j not in used_b:
            used_a.add(i); used_b.add(j); out.append((i, j))
    return sorted(out)
