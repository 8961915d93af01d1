"""The "ATE vs ref" sequence: 72 rendered frames at 192x256 through the headless HIP odometry loop with the parameters of the
reference's config/como.yml.  tests/golden/ate_sequence.npz holds what the reference's own sequential loop produced on the
same frames (tests/golden/make_golden_r2.py::ate_case); used by tests/test_gpu_r2.py and by bench.py's `ate_vs_ref` leg."""
import torch


SEQ640 = {"H": 480, "W": 640, "step": 0.02, "deg": 0.4, "colour": 0, "network_size": [192, 256], "freq_scale": 0.4}
"""The 640x480 sequence of tests/golden/ate_sequence_640.npz (seed 1, 100 frames there) -- bench.py's `odometry_loop` times it."""


def render_frames(G):
    """K, ground-truth poses and the rendered frames (1, 3, H, W) float64 on the CPU of a sequence description (a golden fixture
    or a dict with the same keys): exactly what tests/golden/make_golden_r2.py::ate_frames fed to the reference."""
    from como_amd import synth
    H, W, n, seed = int(G["H"]), int(G["W"]), int(G["nframes"]), int(G["seed"])
    colour = bool(int(G["colour"])) if "colour" in G else False       # `color: rgb`: one texture per channel
    fs = float(G["freq_scale"]) if "freq_scale" in G else W / 640.0
    scenes = [synth.PlaneScene(seed=seed + 1000 * ch, freq_scale=fs) for ch in range(3 if colour else 1)]
    K = synth.intrinsics_for(H, W)
    T = synth.gt_poses(n, step=float(G["step"]), deg=float(G["deg"]))
    g = torch.Generator().manual_seed(seed)
    rgbs = []
    for k in range(n):
        I = torch.stack([sc.render(T[k], K, H, W)[0] for sc in scenes])
        I = I + 0.002 * torch.randn(I.shape, generator=g, dtype=torch.float64)
        rgbs.append(I[None] if colour else I[None].repeat(1, 3, 1, 1))
    return K, T, rgbs


def loop_cfgs(G, pix="float", dev="cuda:0", graph_network=False):
    """Tracking / mapping sections of config/como.yml for the sequence G."""
    H, W = int(G["H"]), int(G["W"])
    colour = bool(int(G["colour"])) if "colour" in G else False
    # network input: the image size in the small fixtures, the reference's fixed 192 x 256 in the 640 x 480 one (Mapping.py:399)
    net = [int(x) for x in G["network_size"]] if "network_size" in G else [H, W]
    tcfg = {"device": dev, "dtype": "float", "color": "rgb" if colour else "gray",
            "pyr": {"start_level": 0, "end_level": 3, "depth_interp_mode": "nearest_neighbor"},
            "term_criteria": {"max_iter": 50, "delta_norm": 1.0e-3, "rel_tol": 1.0e-3, "grad_norm": 1.0},
            "sigmas": {"photo": 1.0e-1},
            "keyframing": {"kf_depth_motion_ratio": 0.12, "kf_num_pixels_frac": 0.75, "one_way_freq": 3}}
    mcfg = {"device": dev, "dtype": "double", "pix_dtype": pix, "color": "rgb" if colour else "gray", "track_ref": {"num_keyframes": 1}, "viewer_snapshots": False,
            "graph": {"num_keyframes": 9, "num_one_way_frames": 24}, "network_size": net, "graph_network": graph_network,
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
    return {"tracking": tcfg, "mapping": mcfg}


KIND_CODE = {None: 0, "init": 3, "keyframe": 1, "one-way": 2}


def run_ate_sequence(G, pix="float", dev="cuda:0"):
    """The headless HIP odometry loop on the sequence of a golden fixture (regenerated from its seeds), with the parameters of
    config/como.yml; returns (kinds, tracked poses dict frame -> (4,4), loop object)."""
    from como_amd import synth
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.odom.sequential import ComoSeq
    K, T, rgbs = render_frames(G)
    model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
    odo = ComoSeq(loop_cfgs(G, pix, dev), K.clone(), (int(G["H"]), int(G["W"])), model)
    kinds, poses = [], {}
    for k in range(len(rgbs)):
        nb = len(odo.est_poses)
        kinds.append(KIND_CODE[odo.iter(1.0 + k, rgbs[k].to(dev))])
        if len(odo.est_poses) > nb:
            poses[k] = odo.est_poses[-1].detach().cpu().double().reshape(4, 4)
    return kinds, poses, odo
