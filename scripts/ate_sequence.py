"""The "ATE vs ref" sequence: 72 rendered frames at 192x256 through the headless HIP odometry loop with the parameters of the
reference's config/como.yml.  tests/golden/ate_sequence.npz holds what the reference's own sequential loop produced on the
same frames (tests/golden/make_golden_r2.py::ate_case); used by tests/test_gpu_r2.py and by bench.py's `ate_vs_ref` leg."""
import torch


def run_ate_sequence(G, pix="float", dev="cuda:0"):
    """The headless HIP odometry loop on the sequence of tests/golden/ate_sequence.npz (regenerated from its seeds), with the
    parameters of config/como.yml; returns (kinds, tracked poses dict frame -> (4,4), loop object)."""
    from como_amd import synth
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.odom.sequential import ComoSeq
    H, W, n, seed = int(G["H"]), int(G["W"]), int(G["nframes"]), int(G["seed"])
    colour = bool(int(G["colour"])) if "colour" in G else False       # `color: rgb`: one texture per channel
    # network input: the image size in the small fixtures, the reference's fixed 192 x 256 in the 640 x 480 one (Mapping.py:399)
    net = [int(x) for x in G["network_size"]] if "network_size" in G else [H, W]
    fs = float(G["freq_scale"]) if "freq_scale" in G else W / 640.0
    scenes = [synth.PlaneScene(seed=seed + 1000 * ch, freq_scale=fs) for ch in range(3 if colour else 1)]
    K = synth.intrinsics_for(H, W)
    T = synth.gt_poses(n, step=float(G["step"]), deg=float(G["deg"]))
    g = torch.Generator().manual_seed(seed)
    tcfg = {"device": dev, "dtype": "float", "color": "rgb" if colour else "gray",
            "pyr": {"start_level": 0, "end_level": 3, "depth_interp_mode": "nearest_neighbor"},
            "term_criteria": {"max_iter": 50, "delta_norm": 1.0e-3, "rel_tol": 1.0e-3, "grad_norm": 1.0},
            "sigmas": {"photo": 1.0e-1},
            "keyframing": {"kf_depth_motion_ratio": 0.12, "kf_num_pixels_frac": 0.75, "one_way_freq": 3}}
    mcfg = {"device": dev, "dtype": "double", "pix_dtype": pix, "color": "rgb" if colour else "gray", "track_ref": {"num_keyframes": 1},
            "graph": {"num_keyframes": 9, "num_one_way_frames": 24}, "network_size": net, "graph_network": False,
            "photo_construction": {"nonmax_suppression_window": 4, "pairwise_batch_size": 128, "radius_thresh": 0.0,
                                   "degrees_thresh": 0.0},
            "term_criteria": {"max_iter": 20, "delta_norm": 1.0e-8, "abs_tol": 1.0e-6, "rel_tol": 1.0e-6},
            "sigmas": {"photo": 1.0e-1, "mean_depth_prior": 1.0e-2, "scale_prior": 1.0e-4, "pose_prior": 1.0e-6},
            "sampling": {"mode": "greedy_conditional_entropy", "max_num_coords": 64, "max_stdev_thresh": 1.0e-2, "border": 3,
                         "fixed_var": 0.0, "dist_thresh": 1.0e-1},
            "corr": {"corr_mode": "logz", "corr_thresh": 3.0e-2, "distill_with_prior": True, "min_obs_depth": 0.0,
                     "logz_grad_mag_thresh": 7.0e-2},
            "init": {"start_level": 0, "end_level": 3, "max_iter": 50, "delta_norm": 1.0e-4, "rel_tol": 1.0e-4,
                     "kf_depth_motion_ratio": 0.04, "kf_num_pixels_frac": 0.75}}
    model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
    odo = ComoSeq({"tracking": tcfg, "mapping": mcfg}, K.clone(), (H, W), model)
    code = {None: 0, "init": 3, "keyframe": 1, "one-way": 2}
    kinds, poses = [], {}
    for k in range(n):
        I = torch.stack([sc.render(T[k], K, H, W)[0] for sc in scenes])
        I = I + 0.002 * torch.randn(I.shape, generator=g, dtype=torch.float64)
        rgb = (I[None] if colour else I[None].repeat(1, 3, 1, 1)).to(dev)
        nb = len(odo.est_poses)
        kinds.append(code[odo.iter(1.0 + k, rgb)])
        if len(odo.est_poses) > nb:
            poses[k] = odo.est_poses[-1].detach().cpu().double().reshape(4, 4)
    return kinds, poses, odo


