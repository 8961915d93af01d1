"""Pair-graph construction (host-side index lists), mirroring the reference's
como/odom/backend/graph_pair_construction.py:155-182.  With the shipped configuration
(radius_thresh = degrees_thresh = 0, config/como.yml:40-41) the graph is: forward + backward
consecutive-keyframe edges, plus temporal one-way edges for recent frames (:97-133).
Radius-based edges (:53-94) are evaluated with small torch ops when the thresholds are positive.
"""
import math

import torch


def get_forward_edges(B):
    return list(range(0, B - 1)), list(range(1, B))


def get_backward_edges(B):
    return list(range(1, B)), list(range(0, B - 1))


def _scaled_dists(p1, p2, med1):
    return torch.cdist(p1[:, :3, 3], p2[:, :3, 3]) / med1[:, None]


def _radius_edges(p1, med1, p2, cfg):
    d = _scaled_dists(p1, p2, med1)
    R12 = p1[:, None, :3, :3].mT @ p2[None, :, :3, :3]
    cos = 0.5 * (R12[..., 0, 0] + R12[..., 1, 1] + R12[..., 2, 2] - 1)
    return (d < cfg["radius_thresh"]) & (cos > math.cos(cfg["degrees_thresh"] * 3.14159 / 180.0)), d


def get_kf_edges(poses, median_depths, cfg):
    ok, _ = _radius_edges(poses, median_depths, poses, cfg)
    i1, i2 = torch.nonzero(ok, as_tuple=True)
    keep = (i1 - i2).abs() > 1
    return i1[keep].tolist(), i2[keep].tolist()


def get_one_way_temporal_neighbors(kf_timestamps, recent_timestamps):
    """Each recent frame is linked to the keyframe before and after it; frames newer than the newest
    keyframe only to that keyframe (reference :97-133)."""
    as_list = lambda ts: [float(t) for t in (ts.tolist() if torch.is_tensor(ts) else ts)]   # ONE read-back for a tensor
    kf_ts, rc_ts = as_list(kf_timestamps), as_list(recent_timestamps)
    nk, nr = len(kf_ts), len(rc_ts)
    kf_ids, r_ids = [], []
    k = -1
    while rc_ts[0] > kf_ts[k + 1]:
        k += 1
        if k == nk - 1:
            break
    j = 0
    if k < nk - 1:
        while j < nr:
            if rc_ts[j] > kf_ts[k + 1]:
                k += 1
            if k >= nk - 1:
                break
            kf_ids += [k, k + 1]
            r_ids += [j, j]
            j += 1
    while j < nr:
        kf_ids.append(k)
        r_ids.append(j)
        j += 1
    return kf_ids, r_ids


def get_one_way_edges(kf_poses, kf_median_depths, recent_poses, kf_timestamps, recent_timestamps, cfg):
    if cfg["radius_thresh"] > 0.0 and cfg["degrees_thresh"] > 0.0:
        ok, d = _radius_edges(kf_poses, kf_median_depths, recent_poses, cfg)
        near = torch.argmin(d, dim=0)
        cols = torch.arange(d.shape[1], device=d.device)
        ok[near, cols] = False
        i1, i2 = torch.nonzero(ok, as_tuple=True)
        return torch.cat((near, i1)).tolist(), torch.cat((cols, i2)).tolist()
    return get_one_way_temporal_neighbors(kf_timestamps, recent_timestamps)


def setup_photometric_pairs(poses, recent_poses, kf_timestamps, recent_timestamps, median_depths, cfg):
    B, R = poses.shape[0], recent_poses.shape[0]
    rf, tf = get_forward_edges(B)
    rb, tb = get_backward_edges(B)
    if cfg["radius_thresh"] > 0.0 and cfg["degrees_thresh"] > 0.0:
        rk, tk = get_kf_edges(poses, median_depths, cfg)
    else:
        rk, tk = [], []
    if R > 0:
        ow_kf, ow_t = get_one_way_edges(poses, median_depths, recent_poses, kf_timestamps, recent_timestamps, cfg)
    else:
        ow_kf, ow_t = [], []
    return rf + rb + rk, tf + tb + tk, ow_kf, ow_t
