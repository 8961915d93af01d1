"""Host-side logic of the product on CPU (no kernels): prior factors, pair graph, landmark bookkeeping, SE3 helpers,
drop-in registration -- against the golden vectors captured from the reference."""
import sys

import pytest
import torch

from tests.conftest import load_golden


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def test_prior_factors_match_reference():
    from como_amd.odom.factors.depth_prior import log_depth_prior
    from como_amd.odom.factors.gp_priors import gp_ml_cost, mean_log_depth_cost
    from como_amd.odom.factors.pixel_prior import pixel_prior_cost
    from como_amd.odom.factors.pose_prior_factors import linearize_pose_prior
    from como_amd.odom.factors.scalar_prior_factors import linearize_multi_scalar_prior, linearize_scalar_prior
    for name in ("ba_window_f64.npz", "ba_window_recent_f64.npz"):
        G = load_golden(name)
        H, g = G["H_photo"].clone(), G["g_photo"].clone()
        kpi = G["kf_inds"][:, :6]
        lmk = G["landmark_inds"]
        dP, dT = G["dlogzm_dzm"] @ G["dzm_dPwm"], G["dlogzm_dzm"] @ G["dzm_dTwc"]
        lm = torch.log(G["median_depths_full"])[:, None, None]
        e = [gp_ml_cost(G["logzm"], lm, G["L_mm"], dP, dT, lmk, kpi, H, g, 1.0)]
        assert rel(H, G["H_gp"]) < 1e-12 and rel(g, G["g_gp"]) < 1e-12
        e.append(log_depth_prior(G["logzm"], lm, dP, dT, G["obs_ref_mask"], lmk, kpi, H, g, "first_mean", 1.0, 1.0))
        assert rel(H, G["H_ld"]) < 1e-12 and rel(g, G["g_ld"]) < 1e-12
        e.append(pixel_prior_cost(G["pm"], G["pm_first_obs"], G["dpm_dPwm"], G["dpm_dTwc"], G["obs_ref_mask"], lmk, kpi, H, g, "first",
                                  1e-2, 3.33e-1))
        assert rel(H, G["H_px"]) < 1e-12 and rel(g, G["g_px"]) < 1e-12
        e.append(linearize_pose_prior(G["kf_poses"][0:1], G["pose_anchor"], H, g, [0, 6], 1e-6))
        e.append(linearize_scalar_prior(G["kf_aff_params"][0, 0:1, :], G["aff_anchor"][0, 0:1, :], H, g, [6, 7], 1e-4))
        e.append(linearize_scalar_prior(G["kf_aff_params"][0, 1:2, :], G["aff_anchor"][0, 1:2, :], H, g, [7, 8], 1e-4))
        L = G["P_m"].shape[0]
        if bool(G["window_full"]):
            fix = G["fix_mask"]
            inds = (torch.arange(3 * L).reshape(L, 3) + int(G["lm_start"]))[fix].reshape(-1)
            e.append(linearize_multi_scalar_prior(G["P_m"][fix].reshape(-1), G["P_anchor"].reshape(-1), H, g, inds, 1e-4))
            assert rel(H, G["H_full"]) < 1e-12 and rel(g, G["g_full"]) < 1e-12
            assert rel(torch.stack([x.reshape(()) for x in e]).double(), G["prior_err"]) < 1e-9


def test_pair_graph_and_landmark_indexing():
    from como_amd.odom.backend.graph_pair_construction import setup_photometric_pairs
    import como_amd.odom.backend.linear_system as ls
    import como_amd.odom.backend.sparse_map as smap
    G = load_golden("ba_window_recent_f64.npz")
    cfg = {"radius_thresh": 0.0, "degrees_thresh": 0.0}
    ref, tgt, owk, owt = setup_photometric_pairs(G["kf_poses"], G["recent_poses"], G["kf_timestamps"], G["recent_timestamps"],
                                                 G["median_depths"], cfg)
    assert ref == G["kf_ref_ids"].tolist() and tgt == G["kf_target_ids"].tolist()
    assert owk == G["ow_kf_ids"].tolist() and owt == G["ow_target_ids"].tolist()
    remap, paired = smap.get_batch_remap_function(G["correspondence_mask"])
    pi = ls.landmark_to_batched_3d_point_inds(paired[0], G["kf_poses"].shape[0])
    assert torch.equal(pi, G["point_inds"])
    out = smap.setup_point_to_frame(G["P_m"], G["kf_poses"], remap, G["intrinsics"], G["P_m"], G["median_depths_in"])
    assert torch.equal(out[0], G["pm"]) and torch.equal(out[1], G["logzm"])           # bit-exact (feeds masks downstream)
    assert rel(out[7], G["dpm_dTwc"]) < 1e-12 and rel(out[5], G["dzm_dTwc"]) < 1e-12
    cn, _ = smap.subselect_pixels(G["kf_img_and_grads"], 2)
    assert torch.equal(cn, G["coords_n"])
    kp, ka, rp, ra, Pn = ls.update_vars(G["delta"], G["kf_poses"], G["kf_aff_params"], G["kf_inds"], G["recent_poses"],
                                        G["recent_aff_params"], G["recent_inds"], G["P_m"], int(G["lm_start"]))
    assert rel(kp, G["kf_poses_new"]) < 1e-12 and rel(rp, G["recent_poses_new"]) < 1e-12 and rel(Pn, G["P_new"]) < 1e-12


def test_setup_test_points_mirror():
    import como_amd.odom.backend.sparse_map as smap
    G = load_golden("ba_window_f64.npz")
    cn = G["coords_n"]
    B = cn.shape[0]
    bi = torch.arange(B)[:, None].expand(-1, cn.shape[1])
    Kt_rows = G["Knm_Kmminv"][bi, cn[..., 0], cn[..., 1], :]
    Pw, dT, dz, med, _, logzn = smap.setup_test_points(G["pm"], G["logzm"], G["kf_poses"], Kt_rows, cn, G["intrinsics"],
                                                       G["dlogzm_dzm"] @ G["dzm_dTwc"], G["dlogzm_dzm"])
    assert torch.equal(Pw, G["Pwn"]) and torch.equal(med, G["median_depths"])
    assert rel(dT, G["dPwn_dTwc"]) < 1e-12 and rel(dz, G["dPwn_dzm"]) < 1e-12


def test_dropin_registration():
    import como_amd
    names = como_amd.install_dropin()
    try:
        import como_backends                                  # noqa: F401  resolves to como_amd.como_backends
        from como.odom.backend.photo import create_photo_system, batch_photo_cost   # noqa: F401
        from como.odom.frontend.photo_tracking import photo_tracking_pyr             # noqa: F401
        assert sys.modules["como_backends"].__name__ == "como_amd.como_backends"
    finally:
        for n in names:
            sys.modules.pop(n, None)


def test_cpu_tensors_are_rejected_loudly():
    import pytest
    import como_amd.como_backends as cb
    x = torch.zeros(1, 2, 2)
    E = torch.eye(2).expand(1, 2, 2, 2).contiguous()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cb.cross_covariance(x, E, x, E, 1.0)


def _cpu_mapping(num_kf=3, m=12):
    from como_amd.odom.Mapping import Mapping
    cfg = {"device": "cpu", "dtype": "double", "color": "gray", "graph": {"num_keyframes": num_kf, "num_one_way_frames": 4},
           "sampling": {"max_num_coords": m}}
    mp = Mapping(cfg, torch.eye(3, dtype=torch.float64))
    mp.init_keyframe_vars()
    mp.init_prior_vals()
    return mp


def test_mapping_window_bookkeeping_vs_reference_snapshots():
    """Correspondence-mask / landmark-list / anchor bookkeeping of Mapping (pure torch) replayed on CPU between consecutive
    snapshots of the reference's run (mapping_f64.npz): every keyframe insertion must reproduce the reference's mask."""
    G = load_golden("mapping_f64.npz")
    tags = [str(t) for t in G["snap_tags"]]
    N = 3
    for i in range(1, len(tags)):
        if not tags[i].startswith("add_kf"):
            continue
        prev, new = G[f"s{i - 1}_correspondence_mask"], G[f"s{i}_correspondence_mask"]
        P_prev, P_new = G[f"s{i - 1}_P_m"], G[f"s{i}_P_m"]
        start = -N + 1
        alive = prev[start:, :].any(dim=0)
        n_alive = int(alive.sum())
        tracked_full = torch.zeros(prev.shape[1], dtype=torch.bool)
        tracked_full[alive] = new[-1, :n_alive]
        corr_mask = tracked_full[prev[-1]]
        mp = _cpu_mapping(N)
        mp.correspondence_mask, mp.P_m = prev.clone(), P_prev.clone()
        mp.initialize_sparse_landmark_vars(corr_mask, P_new[n_alive:])
        assert torch.equal(mp.correspondence_mask, new), tags[i]
        assert torch.equal(mp.P_m, torch.cat((P_prev[alive], P_new[n_alive:]))), tags[i]
        assert bool(mp.window_full) == bool(G[f"s{i}_window_full"]), tags[i]
        if mp.window_full:
            assert torch.equal(mp.P_m_anchors, mp.P_m[new[0]])
    # pose / affine anchors when the window slides: the oldest remaining keyframe becomes the gauge
    mp = _cpu_mapping(N)
    g = torch.Generator().manual_seed(0)
    for k in range(4):
        T = torch.eye(4, dtype=torch.float64)[None].clone()
        T[0, :3, 3] = torch.randn(3, generator=g, dtype=torch.float64)
        mp.initialize_pose_vars(T, torch.randn((1, 2, 1), generator=g, dtype=torch.float64))
    assert mp.kf_poses.shape[0] == N and mp.window_full
    assert torch.equal(mp.pose_anchor, mp.kf_poses[0:1]) and float(mp.kf_aff_params[0].abs().max()) == 0.0
    # one-way frames older than the oldest keyframe are pruned
    mp.kf_timestamps = [5.0, 6.0, 7.0]
    mp.recent_timestamps = [4.0, 4.5, 5.5, 6.5]
    mp.recent_poses = torch.arange(4, dtype=torch.float64).reshape(4, 1, 1).expand(4, 4, 4).clone()
    mp.recent_aff_params = torch.zeros((4, 2, 1), dtype=torch.float64)
    mp.recent_img_and_grads = torch.zeros((4, 3, 2, 2), dtype=torch.float64)
    mp.prune_one_way()
    assert mp.recent_timestamps == [5.5, 6.5] and mp.recent_poses[:, 0, 0].tolist() == [2.0, 3.0]


def test_sliding_predictor_window_equals_cat():
    """Mapping._cat_sliding (the K~ window as a sliding view of one buffer of twice the window's capacity): after every insertion the
    window equals what `torch.cat((old[i:], new))` gives, the kept keyframes are not moved except when the view wraps, and a
    window tensor that is not the view handed out last time (a caller replaced the attribute) is taken over by value."""
    N = 4
    mp = _cpu_mapping(N)
    g = torch.Generator().manual_seed(3)
    ref = torch.empty((0), dtype=torch.float64)
    mp.Knm_Kmminv = torch.empty((0), dtype=torch.float64)
    i = mp.get_kf_start_window_ind()
    moves = 0
    for k in range(23):
        new = torch.randn((1, 5, 6, 3), generator=g, dtype=torch.float64)
        ref = new.clone() if ref.dim() == 1 else torch.cat((ref[i:], new), dim=0)
        old_ptr = mp.Knm_Kmminv[i:].data_ptr() if mp.Knm_Kmminv.dim() > 1 else None
        if k == 11:                                   # a caller swaps in its own tensor: taken over by value
            mp.Knm_Kmminv = mp.Knm_Kmminv.clone()
            old_ptr = None
        mp._cat_sliding("Knm_Kmminv", mp.Knm_Kmminv, new, i)
        assert torch.equal(mp.Knm_Kmminv, ref), k
        assert mp.Knm_Kmminv.shape[0] == min(k + 1, N)
        if old_ptr is not None and mp.Knm_Kmminv.data_ptr() != old_ptr:
            moves += 1
    buf = mp._kt_pp["Knm_Kmminv"]["buf"]
    assert buf.shape[0] == 2 * N
    assert moves <= 23 // N + 1, moves                # the kept keyframes move only when the view reaches the end of the buffer
    with pytest.raises(RuntimeError):
        mp._cat_sliding("Knm_Kmminv", mp.Knm_Kmminv, torch.zeros((2, 5, 6, 3), dtype=torch.float64), -N)
    # a window kept in another element type (the float32 mirror of the float64 predictors): the copy into the window converts
    mp.Knm_Kmminv_pix = torch.empty((0), dtype=torch.float32)
    ref32 = torch.empty((0), dtype=torch.float32)
    for k in range(7):
        new = torch.randn((1, 5, 6, 3), generator=g, dtype=torch.float64)
        ref32 = new.float() if ref32.dim() == 1 else torch.cat((ref32[i:], new.float()), dim=0)
        mp._cat_sliding("Knm_Kmminv_pix", mp.Knm_Kmminv_pix, new, i, dtype=torch.float32)
        assert mp.Knm_Kmminv_pix.dtype == torch.float32 and torch.equal(mp.Knm_Kmminv_pix, ref32), k


def test_sliding_one_way_stack_with_pruned_views_and_its_own_capacity():
    """Round 5: the image stacks live in sliding buffers too -- with the capacity of THEIR window (24 one-way frames, not 9 keyframes)
    and with `prune_one_way` handing back a SUFFIX view of the stack: the next insertion must find the kept frames where they are
    (no copy of the stack) and the window must equal what slicing + `torch.cat` give."""
    N, R = 4, 6
    mp = _cpu_mapping(N)
    g = torch.Generator().manual_seed(9)
    mp.recent_img_and_grads = torch.empty((0), dtype=torch.float64)
    ref = torch.empty((0), dtype=torch.float64)
    i = -R + 1
    moved = 0
    for k in range(40):
        new = torch.randn((1, 3, 5, 7), generator=g, dtype=torch.float64)
        if k in (9, 10, 23) and ref.shape[0] > 2:       # prune_one_way: the two oldest one-way frames leave the window
            ref = ref[2:]
            mp.recent_img_and_grads = mp.recent_img_and_grads[2:]
        ref = new.clone() if ref.dim() == 1 else torch.cat((ref[i:], new), dim=0)
        keep_ptr = mp.recent_img_and_grads[i:].data_ptr() if mp.recent_img_and_grads.dim() > 1 else None
        mp._cat_sliding("recent_img_and_grads", mp.recent_img_and_grads, new, i, cap=R)
        assert torch.equal(mp.recent_img_and_grads, ref), k
        assert mp.recent_img_and_grads.shape[0] <= R
        if keep_ptr is not None and mp.recent_img_and_grads.data_ptr() != keep_ptr:
            moved += 1
    assert mp._kt_pp["recent_img_and_grads"]["buf"].shape[0] == 2 * R
    assert moved <= 40 // R + 1, moved                # the kept frames only move when the view wraps, pruned or not


def test_fill_image_last_point_wins_like_torch_cpu():
    """fill_image's explicit last-wins rule equals what the reference's `img[:, r, c] = vals` does on the CPU."""
    from como_amd.utils.coords import fill_image
    g = torch.Generator().manual_seed(1)
    coords = torch.rand((500, 2), generator=g) * torch.tensor([7.0, 9.0])
    vals = torch.rand((500, 1), generator=g)
    want = float("nan") * torch.ones((1, 7, 9))
    cl = coords.long()
    want[:, cl[..., 0], cl[..., 1]] = vals[..., 0]
    got = fill_image(coords, vals, (7, 9))
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))


def test_tracking_request_rules():
    """Keyframe / one-way request rules of Tracking (Tracking.py:110-161) on hand-made statistics."""
    from como_amd.odom.Tracking import Tracking
    cfg = {"device": "cpu", "dtype": "float", "color": "gray",
           "keyframing": {"kf_depth_motion_ratio": 0.1, "kf_num_pixels_frac": 0.75, "one_way_freq": 3}}
    trk = Tracking(cfg, torch.eye(3), (10, 10))
    trk.init_kf_vars()
    trk.reset_one_way_vars()
    trk.vals_pyr = [torch.zeros((1, 100, 1))]
    T = torch.eye(4)[None].clone()
    med = torch.tensor(2.0)
    T[0, 0, 3] = 0.19
    assert not trk.check_keyframe(med, torch.tensor(90), T)          # 0.19 < 0.1 * 2, 90 % of the pixels still seen
    T[0, 0, 3] = 0.21
    assert trk.check_keyframe(med, torch.tensor(90), T)              # moved far enough
    T[0, 0, 3] = 0.0
    assert trk.check_keyframe(med, torch.tensor(70), T)              # too few pixels re-observed
    trk.last_kf_sent_ts = 5.0                                        # a keyframe request is in flight: no second one
    assert not trk.check_keyframe(med, torch.tensor(10), T)
    # one-way frames: thresholds scaled by (1 + sent + pending) / (1 + freq) = 2/4 while the keyframe is pending
    T[0, 0, 3] = 0.11
    assert trk.check_one_way_frame(med, torch.tensor(100), T, T)     # 0.11 > 0.5 * 0.2
    T[0, 0, 3] = 0.09
    assert not trk.check_one_way_frame(med, torch.tensor(100), T, T)
    assert trk.check_one_way_frame(med, torch.tensor(80), T, T)      # 20 empty pixels > 0.5 * 25


def test_tracking_requests_on_host_scalars_equal_the_tensor_rules():
    """Tracking.decide_frame (the frame graph's one-read-back path) makes the decision check_keyframe / check_one_way_frame make
    on tensors -- random statistics, values ON the thresholds (where the float32 rounding of the products decides), NaN medians."""
    import copy
    from como_amd.odom.Tracking import Tracking
    g = torch.Generator().manual_seed(3)
    n_px = 307200
    for case in range(400):
        ratio = float(torch.rand((), generator=g)) * 0.3 + 0.01
        frac = float(torch.rand((), generator=g)) * 0.8 + 0.1
        cfg = {"device": "cpu", "dtype": "float", "color": "gray",
               "keyframing": {"kf_depth_motion_ratio": ratio, "kf_num_pixels_frac": frac, "one_way_freq": 1 + case % 4}}
        trk = Tracking(cfg, torch.eye(3), (480, 640))
        trk.init_kf_vars()
        trk.reset_one_way_vars()
        trk.vals_pyr = [torch.zeros((1, n_px, 1))]
        trk.num_one_way_since_kf = case % 3
        trk.kf_received_ts = 4.0
        trk.last_kf_sent_ts = 5.0 if case % 5 == 0 else 3.0
        md = torch.rand((), generator=g) * 3 + 0.2
        if case % 37 == 0:
            md = torch.tensor(float("nan"))
        t = torch.randn(3, generator=g) * 0.1
        if case % 3 == 0:                                  # put |t| (almost) exactly on one of the two motion thresholds
            pending = 1 if trk.last_kf_sent_ts > trk.kf_received_ts else 0
            sc = 1.0 if case % 2 else (1.0 + trk.num_one_way_since_kf + pending) / (1.0 + cfg["keyframing"]["one_way_freq"])
            t = t / torch.linalg.norm(t) * (torch.tensor(sc * ratio, dtype=torch.float32) * md)
        nseen = int(torch.randint(0, n_px + 1, (), generator=g))
        if case % 4 == 1:                                  # ... or the pixel count on the keyframe threshold
            nseen = int(frac * n_px) + (case % 3) - 1
        T = torch.eye(4)[None].clone()
        T[0, :3, 3] = t
        ref = copy.copy(trk)
        if ref.check_keyframe(md, torch.tensor(nseen, dtype=torch.int32), T):
            want = "keyframe"
        elif ref.check_one_way_frame(md, torch.tensor(nseen, dtype=torch.int32), T, T):
            want = "one-way"
        else:
            want = None
        got = trk.decide_frame(float(torch.linalg.norm(T[:, :3, 3])), float(md), nseen, T)
        assert got == want, (case, got, want)
        assert int(trk.last_one_way_empty_pixels) == int(ref.last_one_way_empty_pixels)


def test_persistent_pyramid_buffers_skip_copies_of_their_own_views():
    """_PyrBuffers.load_reference: arrays that already live in the buffers (the tracker's reference kernels write there directly and
    hand views back) are not copied; foreign arrays are; an unchanged tuple of sources is recognised by identity."""
    from como_amd.odom.frontend.photo_tracking import _PyrBuffers
    pb = _PyrBuffers.from_shapes(1, 1, [(3, 4), (6, 8)], "cpu", torch.float32)
    assert [tuple(c["P"].shape) for c in pb.levels] == [(1, 12, 3), (1, 48, 3)] and pb.levels[1]["dI"].shape == (1, 48, 1, 8)
    g = torch.Generator().manual_seed(0)
    for c in pb.levels:
        for k in ("vals", "P", "dI"):
            c[k].copy_(torch.rand(c[k].shape, generator=g))
        c["mask"].fill_(1)
    before = [{k: c[k].clone() for k in ("vals", "P", "dI", "mask")} for c in pb.levels]
    views = lambda k, shp: [c[k].view(shp(c[k])) for c in pb.levels]
    vals = views("vals", lambda t: (1, t.shape[1], 1))
    P = views("P", lambda t: (1, t.shape[1], 3))
    dI = views("dI", lambda t: (1, t.shape[1], 1, 8))
    masks = [c["mask"].view(torch.bool).view(1, -1) for c in pb.levels]
    pb.load_reference(vals, P, dI, masks)                               # all aliases: nothing may change
    for c, b in zip(pb.levels, before):
        assert all(torch.equal(c[k], b[k]) for k in b)
    P2 = [torch.rand(p.shape, generator=g) for p in P]                  # foreign points: copied in, the rest untouched
    pb.load_reference(vals, P2, dI, masks)
    for c, b, p2 in zip(pb.levels, before, P2):
        assert torch.equal(c["P"], p2.reshape(c["P"].shape)) and torch.equal(c["vals"], b["vals"]) and torch.equal(c["dI"], b["dI"])
    P2[0].add_(1.0)                                                     # same tensor objects again: recognised, not re-read
    pb.load_reference(vals, P2, dI, masks)
    assert not torch.equal(pb.levels[0]["P"], P2[0].reshape(pb.levels[0]["P"].shape))


def test_trajectory_io_roundtrip(tmp_path):
    """save_traj writes TUM lines (timestamp tx ty tz qx qy qz qw, 4 decimals) that read back to the poses."""
    import numpy as np
    from como_amd.geometry.lie_algebra import se3_exp
    from como_amd.utils.io import pose_to_tq, save_traj, tq_to_pose
    g = torch.Generator().manual_seed(3)
    T = se3_exp(0.3 * torch.randn((5, 6), generator=g, dtype=torch.float64))
    tq = pose_to_tq(T)
    assert tq.shape == (5, 7) and np.allclose(np.linalg.norm(tq[:, 3:], axis=1), 1.0)
    assert np.allclose(tq_to_pose(tq), T.numpy(), atol=1e-12)
    assert pose_to_tq(T[0]).shape == (7,)
    p = tmp_path / "traj.txt"
    save_traj(str(p), [0.5 * k for k in range(5)], T)
    rows = np.loadtxt(str(p))
    assert rows.shape == (5, 8) and np.allclose(rows[:, 0], [0.0, 0.5, 1.0, 1.5, 2.0])
    assert np.allclose(tq_to_pose(rows[:, 1:]), T.numpy(), atol=2e-4)              # 4 decimals


def test_slabwise_gram_equals_direct():
    """distill_depth._gram (256-slab batched GEMM for tall least-squares systems) == A^T A, A^T b."""
    from como_amd.depth_cov.core.distill_depth import _gram
    g = torch.Generator().manual_seed(5)
    for n in (100, 4096, 10007):
        A = torch.randn((1, n, 7), generator=g, dtype=torch.float64)
        b = torch.randn((1, n, 1), generator=g, dtype=torch.float64)
        AtA, Atb = _gram(A, b)
        assert rel(AtA, A.mT @ A) < 1e-13 and rel(Atb, A.mT @ b) < 1e-12
    # (the solve of the normal equations runs in csrc/smallsolve.hip: tests/test_gpu_r3.py::test_small_spd_conditioning)


def test_pair_graph_radius_and_degree_edges_vs_reference():
    """setup_photometric_pairs with positive radius / degree thresholds (graph_pair_construction.py:21-94,155-182): keyframe
    radius edges and pose-based one-way edges, against pair lists the reference produced (make_golden_r2.py::pair_graph_case)."""
    from como_amd.odom.backend.graph_pair_construction import setup_photometric_pairs
    G = load_golden("pair_graph.npz")
    n_extra = 0
    for i in range(4):
        rad, deg = (float(x) for x in G[f"c{i}_cfg"])
        a, b, c, d = setup_photometric_pairs(G["kf_poses"], G["recent_poses"], G["kf_timestamps"], G["recent_timestamps"],
                                             G["median_depths"], {"radius_thresh": rad, "degrees_thresh": deg})
        assert a == G[f"c{i}_kf_ref"].tolist() and b == G[f"c{i}_kf_tgt"].tolist(), i
        assert c == G[f"c{i}_ow_kf"].tolist() and d == G[f"c{i}_ow_tgt"].tolist(), i
        n_extra += len(a) - 12
    assert n_extra > 0                                                # the radius edges are actually exercised


def test_pair_table_colour_expansion():
    """`color: rgb`: every keyframe pair becomes c consecutive entries of the pair table, one per channel, with the same system
    rows; two-pair groups never mix reference keyframes or channels (the block kernel loads I_ref once per group)."""
    import torch
    from como_amd.odom.backend.photo import PairTable
    B, m, L, HW = 4, 8, 20, 48 * 64
    kf_inds = torch.arange(8 * B).reshape(B, 8)
    recent_inds = torch.arange(8 * 2).reshape(2, 8) + 8 * B
    lm = torch.arange(3 * m * B).reshape(B, 3 * m) + 100
    ref = [0, 1, 2, 1, 2, 3, 1, 2]
    tgt = [1, 2, 3, 0, 1, 2, 0, 1]
    rec = [False] * 6 + [True, True]
    for c in (1, 3):
        stack = 3 * c * HW
        t = PairTable(ref, tgt, rec, B, kf_inds, recent_inds, lm, stack, B * stack, torch.device("cpu"), channels=c)
        assert t.b == len(ref) * c and t.npairs == len(ref) and t.channels == c
        if c == 1:
            assert t.pair_chan is None
        else:
            assert t.pair_chan.tolist() == [0, 1, 2] * len(ref)
        for p in range(len(ref)):
            for ch in range(c):
                e = p * c + ch
                assert int(t.ref_slot[e]) == ref[p]
                frame = tgt[p] + (B if rec[p] else 0)
                assert int(t.tgt_pose[e]) == frame and int(t.tgt_aff[e]) == frame
                assert int(t.tgt_img[e]) == (B * stack + tgt[p] * stack if rec[p] else tgt[p] * stack)   # the FRAME's stack
                assert torch.equal(t.pose_ref_inds[e], kf_inds[ref[p]]) and torch.equal(t.landmark_inds[e], lm[ref[p]])
                assert torch.equal(t.pose_tgt_inds[e], (recent_inds[tgt[p]] if rec[p] else kf_inds[tgt[p]]))
        seen = []
        for a, b_ in t.grp_pairs.tolist():
            seen.append(a)
            if b_ >= 0:
                seen.append(b_)
                assert int(t.ref_slot[a]) == int(t.ref_slot[b_]) and a % c == b_ % c       # same keyframe, same channel
        assert sorted(seen) == list(range(t.b))                                          # every entry exactly once


def test_window_state_lives_in_one_buffer_and_snapshot_is_a_copy():
    """WindowBA keeps poses | affine parameters | landmarks | median depths in ONE buffer (`state_flat`; the kernels get pointers into
    it) so that the sequential loop publishes the state with one copy: the named tensors are views of it at 16-byte steps, and
    `snapshot_state` returns equal but independent tensors."""
    import copy
    from como_amd import synth
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    B, m = 3, 16
    eye = torch.eye(m, dtype=torch.float64)[None].repeat(B, 1, 1)
    g = torch.Generator().manual_seed(0)
    st = synth.make_window(B=B, H=48, W=64, m=m, dtype=torch.float64, device="cpu", seed=1,
                           predictor=lambda cov, cm: (eye, eye, torch.rand((B, 48, 64, m), generator=g, dtype=torch.float64)))
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 4
    w = WindowBA(st, cfg=cfg, pix_dtype=torch.float64, window_full=True, fused=False)
    base = w.state_flat.data_ptr()
    for t in (w.poses_all, w.aff_all, w.P_m, w.median_depths):
        off = t.data_ptr() - base
        assert 0 <= off < w.state_flat.numel() * 8 and off % 16 == 0 and t.is_contiguous()
    assert torch.equal(w.P_m, st["P_m"]) and w.P_m.data_ptr() != st["P_m"].data_ptr()        # (a copy: the caller's tensor is not updated in place)
    assert torch.equal(w.kf_poses, st["kf_poses"]) and torch.equal(w.median_depths, st["median_depth_init"])
    sn = w.snapshot_state()
    assert torch.equal(sn["poses"], w.poses_all) and torch.equal(sn["aff"], w.aff_all) and torch.equal(sn["P_m"], w.P_m)
    w.poses_all[0, 0, 3] += 1.0
    w.P_m[0, 0] -= 1.0
    assert float(w.state_flat[3]) == float(w.poses_all[0, 0, 3]) and not torch.equal(sn["poses"], w.poses_all) and not torch.equal(sn["P_m"], w.P_m)
    assert sn["poses"].dtype == torch.float64 and w.snapshot_state(torch.float32)["poses"].dtype == torch.float32


def test_cached_coordinate_helpers_and_pose_composition_equal_the_plain_forms():
    """Host-side helpers of the keyframe path: the cached normalisation constants / pixel grids return what the uncached
    expressions give (reference como/utils/coords.py:12-15, 31-36), and `composeSE3`'s torch form (CPU tensors; the GPU path is
    `como_se3_compose_*`) equals get_rel_pose / get_T_w_curr as the reference writes them (transforms.py:6-13)."""
    from como_amd import synth
    from como_amd.geometry.lie_algebra import composeSE3, invertSE3
    from como_amd.geometry.transforms import get_rel_pose, get_T_w_curr
    from como_amd.utils.coords import get_test_coords, normalize_coordinates
    g = torch.Generator().manual_seed(5)
    x = 200.0 * torch.rand((1, 37, 2), generator=g, dtype=torch.float64)
    for dims in ((192, 256), torch.Size((480, 640)), [48, 64]):
        A = 1.0 / torch.as_tensor([float(d) for d in dims], dtype=torch.float64)
        ref = 2 * A * x + A - 1
        assert torch.equal(normalize_coordinates(x, dims), ref) and torch.equal(normalize_coordinates(x, dims), ref)   # (second call: cached)
        assert torch.equal(normalize_coordinates(x.float(), dims), 2 * A.float() * x.float() + A.float() - 1)
    c1, c2 = get_test_coords((5, 7), "cpu"), get_test_coords((5, 7), "cpu")
    assert c1 is c2 and c1.shape == (1, 35, 2) and c1[0, 8].tolist() == [1, 1] and c1[0, -1].tolist() == [4, 6]
    assert get_test_coords((5, 7), "cpu", batch_size=2).shape == (2, 35, 2)
    A = synth.se3_exp(0.3 * torch.randn((4, 6), generator=g, dtype=torch.float64))
    B = synth.se3_exp(0.3 * torch.randn((4, 6), generator=g, dtype=torch.float64))
    assert torch.equal(composeSE3(A, B, 1), invertSE3(A) @ B) and torch.equal(composeSE3(A[:1], B, 2), A[:1] @ invertSE3(B))
    assert torch.equal(get_rel_pose(A, B), invertSE3(A) @ B) and torch.equal(get_T_w_curr(A, B), A @ invertSE3(B))
    assert float((get_rel_pose(A, A) - torch.eye(4, dtype=torch.float64)).abs().max()) < 1e-14


def test_cached_index_ramp_and_initialiser_statics():
    """Round 5 host helpers: `window_ba._ar` hands out views of ONE cached ramp (equal to torch.arange, also after it had to grow);
    `two_frame_sfm.photo_statics` holds exactly what `construct_photo_system` used to build in every Gauss-Newton iteration
    (reference two_frame_sfm.py:232-269: pixel index row * W + col, intensities as (1,N,c), ray = ((col - cx) / fx, (row - cy) / fy, 1))."""
    from como_amd.odom import window_ba as wb
    a = wb._ar(37, "cpu")
    assert torch.equal(a, torch.arange(37)) and a.dtype == torch.long
    big = wb._ar(10000, "cpu")                                      # beyond the first allocation: a new ramp, the old view stays valid
    assert torch.equal(big, torch.arange(10000)) and torch.equal(a, torch.arange(37))
    assert wb._ar(5, "cpu").data_ptr() == wb._ar(9, "cpu").data_ptr()
    assert torch.equal(wb._ar(6, "cpu", torch.int32), torch.arange(6, dtype=torch.int32))
    assert torch.equal(wb._ar(24, "cpu")[8:].reshape(2, 8), torch.arange(16).reshape(2, 8) + 8)

    from como_amd.odom.frontend.two_frame_sfm import photo_statics
    g = torch.Generator().manual_seed(2)
    h, w, m, c = 6, 8, 4, 3
    N = h * w
    rows, cols = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    coords = torch.stack((rows.reshape(-1), cols.reshape(-1)), dim=-1)[None]
    vals = torch.rand((1, c, N), generator=g, dtype=torch.float64)
    Kt = torch.rand((1, N, m), generator=g, dtype=torch.float64)
    img = torch.rand((1, 3 * c, h, w), generator=g, dtype=torch.float64)
    K = torch.tensor([[5.0, 0, 3.5], [0, 4.0, 2.5], [0, 0, 1]], dtype=torch.float64)
    st = photo_statics(coords, vals, Kt, img, K)
    assert torch.equal(st["pixcoord"], torch.arange(N, dtype=torch.int32)[None])
    assert st["vals"].shape == (1, N, c) and torch.equal(st["vals"][0, :, 1], vals[0, 1])
    assert torch.equal(st["poses"][1], torch.eye(4, dtype=torch.float64)) and st["poses"].shape == (2, 4, 4)
    ray = st["ray"]
    assert ray.shape == (1, N, 3) and torch.equal(ray[0, :, 2], torch.ones(N, dtype=torch.float64))
    k = 2 * w + 5                                                   # pixel (row 2, col 5)
    assert ray[0, k, 0].item() == (5 - 3.5) / 5.0 and ray[0, k, 1].item() == (2 - 2.5) / 4.0
    assert st["Kt"].data_ptr() == Kt.data_ptr() and st["img"].data_ptr() == img.data_ptr()      # (already in the system's type: no copies)
