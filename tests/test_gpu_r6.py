"""Round-6 pins of the HIP path (through the C ABI), `-m gpu` on an MI355X:
  * the two spin-wait kernels fail LOUDLY and recoverably: the XCD-local tracking level checks its own placement inside the
    launch (status -2) and the host falls back; the persistent Cholesky's time-out (info -1) and a non-positive pivot never
    reach the state (device-side guard of the update) and are acted on by `WindowBA.check_solver` / `Mapping._check_solver`
    (the reference swallows the status: como/odom/backend/linear_system.py:109);
  * (further down) the round's new kernels against the forms they replace."""
import copy

import pytest
import torch

from tests.conftest import report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ---------------------------------------------------------------------------------------------------------------------
def test_xcd_local_level_checks_its_own_placement():
    """csrc/track.hip: every workgroup of an XCD-local level adds itself to a per-XCD census (HW_REG_XCC_ID) and workgroup 0 reports
    status -2 unless all of them share one XCD.  (i) an ordinary launch passes the census (status 0); (ii) with the debug switch
    one workgroup reports a neighbouring XCD: the level's record carries -2, `photo_tracking_pyr` discards the result, tracks
    the frame on the per-iteration chain (the same pose as an undisturbed run) and switches the local form off for the process:
    the next level launch runs device-wide with status 0."""
    import como_amd.odom.frontend.photo_tracking as pt
    from como_amd import _lib
    from tests.test_gpu_r2 import _tracking_level_inputs
    L = _lib.lib()
    if L.como_track_level_probe() != 1:
        pytest.skip("dispatcher placement probe says no XCD-local form on this device")
    prev = L.como_track_level_set_local(1)
    try:
        H, W = 60, 80
        tp, K, P, vals, J = _tracking_level_inputs(H, W, 3)
        aff = torch.zeros((1, 2, 1), device=DEV)
        mask = torch.ones(P.shape[1], dtype=torch.uint8, device=DEV)
        term = {"max_iter": 50, "delta_norm": 1e-3, "rel_tol": 1e-3, "grad_norm": 1.0}
        assert L.como_track_level_local_state() == 1
        T0, a0 = pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J.clone(), tp["img_cur"], K, 0.1, term, in_mask=mask, fused=True)
        rec0 = pt.photo_level_tracking.last_out.cpu()
        assert int(rec0[104]) == 0 and int(rec0[105]) >= 1
        # --- forced mismatch: the level itself reports it
        L.como_track_level_debug_mismatch(1)
        pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J.clone(), tp["img_cur"], K, 0.1, term, in_mask=mask, fused=True)
        rec1 = pt.photo_level_tracking.last_out.cpu()
        assert int(rec1[104]) == -2
        # --- the pyramid entry point acts on it
        fb0 = getattr(pt.photo_tracking_pyr, "fallbacks", 0)
        Tp, ap = pt.photo_tracking_pyr(tp["Tji_init"], aff, [vals], [P], [J.clone()], [mask.bool()], [K], [tp["img_cur"]], 0.1, term)
        assert getattr(pt.photo_tracking_pyr, "fallbacks", 0) == fb0 + 1
        assert L.como_track_level_local_state() == 0
        L.como_track_level_debug_mismatch(0)
        Tc, ac = pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J.clone(), tp["img_cur"], K, 0.1, term, in_mask=mask, fused=False)
        eT = (Tp - Tc).abs().max().item()
        # --- and the device-wide form serves the same level from now on
        T2, a2 = pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J.clone(), tp["img_cur"], K, 0.1, term, in_mask=mask, fused=True)
        rec2 = pt.photo_level_tracking.last_out.cpu()
        report("xcd_census", status_ok=int(rec0[104]), status_forced=int(rec1[104]), status_after=int(rec2[104]),
               fallback_vs_chain=eT, wide_vs_local=(T2 - T0).abs().max().item())
        assert int(rec2[104]) == 0 and int(rec2[105]) == int(rec0[105])
        assert eT < 1e-7
        assert (T2 - T0).abs().max().item() < 2e-6
    finally:
        L.como_track_level_debug_mismatch(0)
        L.como_track_level_set_local(prev)


def _small_window(seed=5, B=3, H=96, W=128, m=16):
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    st = synth.make_window(B=B, H=H, W=W, m=m, dtype=torch.float64, device=DEV, seed=seed,
                           predictor=lambda cov, cm: prep_predictor(cov, cm, 1.0))
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 2
    return WindowBA(st, cfg=cfg, pix_dtype=torch.float64, window_full=True)


def test_solver_status_guards_the_update():
    """(i) como_win_update_checked leaves poses / affine parameters / landmarks alone when the status word is non-zero;
    (ii) the persistent solver's time-out (forced: its chain workgroup leaves at once, the others' bounded waits expire) ends with
    info = -1, the window's state is bit-for-bit what it was, `check_solver` switches to the multi-launch solver, redoes the
    iteration and lands where an undisturbed window lands; (iii) a non-finite system (a NaN landmark -> H[0][0] poisoned) raises
    instead of updating."""
    from como_amd import _lib
    L = _lib.lib()
    prev = L.como_chol_set_persistent(1)
    try:
        ref = _small_window()
        d_ref = ref.iterate().clone()
        assert ref.check_solver() is True
        # (i)
        wb = _small_window()
        s0 = wb.state_flat.clone()
        one = torch.ones(1, dtype=torch.int32, device=DEV)
        rc = L.como_win_update_checked(d_ref.data_ptr(), wb.poses_all.data_ptr(), wb.aff_all.data_ptr(), wb.frame_inds.data_ptr(), wb.F,
                                       wb.P_m.data_ptr(), wb.L, wb.lm_start, one.data_ptr(), _lib.stream_ptr(torch.device(DEV)))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(wb.state_flat, s0)
        # (ii)
        persistent = L.como_chol_persistent_state() == 1
        if persistent:
            L.como_chol_debug_stall(1)
            wb.iterate()
            L.como_chol_debug_stall(0)
            torch.cuda.synchronize()
            assert int(wb.info.item()) == -1
            # (the scaffold may re-initialise landmarks and the priors store median depths before the solve: compare poses / affine)
            assert torch.equal(wb.poses_all.reshape(-1), s0[:16 * wb.F])
            assert wb.check_solver() is False
            assert L.como_chol_persistent_state() == 0
            # the undisturbed twin: the timed-out pass had linearised once (median depths / re-initialised landmarks stored)
            twin = _small_window()
            twin.linearize()
            twin.iterate()
            e = (wb.poses_all - twin.poses_all).abs().max().item()
            report("solver_timeout", info_after=int(wb.info.item()), pose_diff_vs_undisturbed=e)
            assert int(wb.info.item()) == 0 and e < 1e-9
            L.como_chol_set_persistent(1)
        # (iii)
        bad = _small_window()
        p0 = bad.poses_all.clone()
        bad.P_m[0, 0] = float("nan")
        bad.iterate()
        torch.cuda.synchronize()
        assert int(bad.info.item()) > 0
        assert torch.equal(bad.poses_all, p0)
        with pytest.raises(RuntimeError, match="not positive definite"):
            bad.check_solver()
        report("solver_status", persistent=persistent, info_nan=int(bad.info.item()))
    finally:
        L.como_chol_debug_stall(0)
        L.como_chol_set_persistent(prev)


# ---------------------------------------------------------------------------------------------------------------------
def _write_tum_tree(root, n, step=0.02, deg=0.4, seed=1):
    """A TUM-format sequence (`rgb.txt` + `rgb/<timestamp>.png`, freiburg3 = no rectification) rendered from the plane scene of the
    pinned loop fixtures at the camera's native 480x640 with the freiburg3 intrinsics; returns (sequence path, GT poses)."""
    import numpy as np
    from PIL import Image
    from como_amd import synth
    from como_amd.data.odom_datasets import TumOdometryDataset
    seq = root / "tum" / "rgbd_dataset_freiburg3_synthetic_plane"
    (seq / "rgb").mkdir(parents=True)
    K = torch.tensor(TumOdometryDataset.CAMERAS[3][0], dtype=torch.float64)
    sc = synth.PlaneScene(seed=seed, freq_scale=0.4)
    T = synth.gt_poses(n, step=step, deg=deg)
    lines = ["# color images\n", "# file: synthetic\n", "# timestamp filename\n"]
    for k in range(n):
        I = sc.render(T[k], K, 480, 640)[0].clamp(0, 1)
        u8 = (I * 255.0 + 0.5).to(torch.uint8).numpy()
        ts = "%.6f" % (1000.0 + k / 30.0)
        Image.fromarray(np.repeat(u8[..., None], 3, axis=2)).save(str(seq / "rgb" / (ts + ".png")))
        lines.append(f"{ts} rgb/{ts}.png\n")
    (seq / "rgb.txt").write_text("".join(lines))
    return str(seq) + "/", T


def test_headless_runner_on_a_tum_tree(tmp_path):
    """`python -m como_amd.run` (como/como_dataset.py:11-37 + GuiWindow.update_main :528-599 + save_traj :359-367 without the
    GUI): YAML -> dataset reader -> ComoSeq.iter per frame -> keyframe trajectory in TUM format.  The file equals what driving the
    same pieces by hand writes (same frames, same config), the first frame never reaches the loop (as in the reference), the
    initialisation completes, and the tracked trajectory follows the rendered camera path (sim(3)-aligned ATE)."""
    import os
    import numpy as np
    from como_amd import run
    from como_amd.utils.ate import ate_rmse
    from como_amd.utils.io import tq_to_pose
    n = 40
    seq, T_gt = _write_tum_tree(tmp_path, n)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "results" / "traj.txt")
    argv = ["--dataset_type", "tum", "--dataset_dir", seq, "--config", os.path.join(root, "config", "como.yml"), "--out", out,
            "--device", DEV, "--random_weights", "0", "--pix_dtype", "float"]
    assert run.main(argv) == out
    rows = np.loadtxt(out).reshape(-1, 8)
    frames = np.loadtxt(out[:-4] + "_frames.txt").reshape(-1, 8)
    # by hand
    ds = run.get_dataset("tum", [192, 256], seq)
    cfg = run.load_slam_cfg(os.path.join(root, "config", "como.yml"), DEV, "float")
    odo, hist, kinds = run.run_sequence(ds, cfg, run.load_model(cfg["mapping"], DEV, 0), first_frame=1)
    assert len(kinds) == n - 1 and kinds[0] == "init"
    assert rows.shape[0] == len(hist.timestamps) >= 3
    assert np.allclose(rows[:, 0], np.round(np.array(hist.timestamps), 4))
    want = torch.stack(hist.poses).numpy()
    got = tq_to_pose(rows[:, 1:])
    assert np.abs(got[:, :3, 3] - want[:, :3, 3]).max() < 1.01e-4            # (four decimals in the file)
    assert rows[0, 0] == pytest.approx(round(1000.0 + 1 / 30.0, 4))          # keyframe 0 = frame 1: frame 0 is the renderer's
    # the tracked poses against the rendered camera path
    idx = [int(round((t - 1000.0) * 30.0)) for t in frames[:, 0]]
    est = [torch.from_numpy(tq_to_pose(r[1:])) for r in frames]
    gt = [T_gt[i] for i in idx]
    ate = ate_rmse(est, gt, "sim3")
    path = float(sum((T_gt[k + 1, :3, 3] - T_gt[k, :3, 3]).norm() for k in range(n - 1)))
    report("headless_runner", frames=len(kinds), keyframes=int(rows.shape[0]), tracked=len(est), ate_sim3_vs_gt=ate, path_length=path,
           kinds={str(k): kinds.count(k) for k in set(kinds)})
    assert len(est) >= n - 6 and ate < 0.02 * path


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rgb_dtype", [torch.float32, torch.float64])
def test_frame_handover_kernels_equal_the_torch_chains(rgb_dtype):
    """The one-launch forms of a frame hand-over (tracker -> mapper, Mapping.py:580-598 / :369-379) against the chains they replace,
    BIT for bit: como_frame_stack_f64 = rgb.to(float64) -> rgb_to_grayscale -> ImageGradientModule -> cat (+ the float32 mirror's
    rounding); como_frame_world_f64 = get_T_w_curr (transforms.py:6-8) and get_aff_w_curr (affine_brightness.py:5-10) on the
    widened values."""
    from como_amd import _lib, synth
    from como_amd.geometry.affine_brightness import get_aff_w_curr
    from como_amd.geometry.transforms import get_T_w_curr
    from como_amd.utils.image_processing import img_and_grads, rgb_to_grayscale
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    H, W = 96, 130
    rgb = torch.rand((1, 3, H, W), generator=g, dtype=torch.float64).to(rgb_dtype).to(DEV)
    want = img_and_grads(rgb_to_grayscale(rgb.to(torch.float64)))
    stack = torch.empty((1, 3, H, W), dtype=torch.float64, device=DEV)
    pix = torch.empty((1, 3, H, W), dtype=torch.float32, device=DEV)
    s = _lib.stream_ptr(torch.device(DEV))
    assert L.como_frame_stack_f64(rgb.data_ptr(), 1 if rgb_dtype == torch.float32 else 0, H, W, stack.data_ptr(), pix.data_ptr(), s) == 0
    assert torch.equal(stack, want) and torch.equal(pix, want.float())
    stack2 = torch.zeros_like(stack)
    assert L.como_frame_stack_f64(rgb.data_ptr(), 1 if rgb_dtype == torch.float32 else 0, H, W, stack2.data_ptr(), None, s) == 0
    assert torch.equal(stack2, want)
    T_w_kf = synth.se3_exp(0.3 * torch.randn((1, 6), generator=g, dtype=torch.float64)).to(DEV)
    T_c = synth.se3_exp(0.05 * torch.randn((1, 6), generator=g, dtype=torch.float64)).to(rgb_dtype).to(DEV)
    a_kf = (0.1 * torch.randn((1, 2, 1), generator=g, dtype=torch.float64)).to(DEV)
    a_c = (0.1 * torch.randn((1, 2, 1), generator=g, dtype=torch.float64)).to(rgb_dtype).to(DEV)
    out = torch.empty(18, dtype=torch.float64, device=DEV)
    assert L.como_frame_world_f64(T_w_kf.data_ptr(), T_c.data_ptr(), a_kf.data_ptr(), a_c.data_ptr(), 1 if rgb_dtype == torch.float32 else 0,
                                  out.data_ptr(), out[16:].data_ptr(), s) == 0
    Tw = get_T_w_curr(T_w_kf, T_c.to(torch.float64))
    aw = get_aff_w_curr(a_kf, a_c.to(torch.float64))
    assert torch.equal(out[:16].view(1, 4, 4), Tw) and torch.equal(out[16:].view(1, 2, 1), aw)


@pytest.mark.parametrize("speculate", [None, "hit", "miss"])
def test_retargeted_window_equals_a_rebuilt_one(speculate):
    """(speculate: the pair table of the re-targeted topology built ahead by `WindowBA.speculate` -- what the sequential loop does
    while the tracker runs -- for the right / for another set of one-way frames: adopted / ignored, same result either way.)
    WindowBA.retarget (same keyframes, another set of one-way frames -- what 55 % of the sequential loop's frames ask for) against
    a window built from scratch on the same state: the pair lists, the system size and one Gauss-Newton iteration's result
    (poses / affine parameters of every frame, landmarks) are identical, bit for bit -- the normal equations are assembled in
    fixed point, so nothing depends on which buffers were reused."""
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    B, H, W, m = 3, 96, 128, 16
    st = synth.make_window(B=B, H=H, W=W, m=m, dtype=torch.float64, device=DEV, seed=7,
                           predictor=lambda cov, cm: prep_predictor(cov, cm, 1.0))
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 2
    def with_recent(k):
        s2 = dict(st)
        if k:
            s2.update(synth.make_recent([0.3, 1.3, 1.6, 0.6, 1.8][:k], H, W, 7, device=DEV))
        return s2

    s1, s3 = with_recent(1), with_recent(3)
    if speculate:
        s1["corr_host"] = s3["corr_host"] = st["correspondence_mask"].cpu().numpy()      # (the host mirror Mapping keeps)
    wb = WindowBA(s1, cfg=cfg, pix_dtype=torch.float64, window_full=True, rec_capacity=4)
    wb.iterate()
    # the caller's state after that iteration (what Mapping.iterate publishes), then two more one-way frames
    sn = wb.snapshot_state()
    for s_ in (s3,):
        s_["kf_poses"], s_["kf_aff_params"] = sn["poses"][:B].clone(), sn["aff"][:B].reshape(B, 2, 1).clone()
        s_["P_m"], s_["median_depth_init"] = sn["P_m"].clone(), sn["median"].clone()
        s_["recent_poses"] = torch.cat((sn["poses"][B:B + 1], s_["recent_poses"][1:]))
    fresh = WindowBA(s3, cfg=cfg, pix_dtype=torch.float64, window_full=True)
    if speculate:
        assert getattr(wb, "_host_np", None) is not None
        rec_ts = s3["recent_timestamps"].tolist()
        wb.speculate(s3["kf_timestamps"].tolist(), rec_ts if speculate == "hit" else rec_ts[:2],
                     s3["recent_img_and_grads"].data_ptr())
        assert wb._spec is not None
    assert wb.retarget(s3) is True
    assert getattr(wb, "spec_hits", 0) == (1 if speculate == "hit" else 0) and getattr(wb, "_spec", None) is None
    assert wb.F == fresh.F == B + 3 and wb.dim == fresh.dim
    assert (wb.kf_pairs, wb.one_way_pairs) == (fresh.kf_pairs, fresh.one_way_pairs)
    d1, d2 = wb.iterate().clone(), fresh.iterate().clone()
    assert int(wb.info.item()) == 0 and int(fresh.info.item()) == 0
    report("retarget", dim=wb.dim, delta_diff=(d1 - d2).abs().max().item(), pose_diff=(wb.poses_all - fresh.poses_all).abs().max().item())
    assert torch.equal(wb.H, fresh.H) and torch.equal(wb.g, fresh.g)
    assert torch.equal(d1, d2) and torch.equal(wb.poses_all, fresh.poses_all) and torch.equal(wb.P_m, fresh.P_m)
    # a state this window cannot take: more one-way frames than its capacity, another keyframe set
    assert wb.retarget(with_recent(5)) is False
    other = dict(s3)
    other["kf_img_and_grads"] = st["kf_img_and_grads"].clone()
    assert wb.retarget(other) is False


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(240, 320), (480, 640)])
def test_persistent_sampler_loop_gives_the_same_picks(h, w, monkeypatch):
    """como_greedy_persist_f32 (csrc/cov.hip greedy_persist_kernel: the large-domain greedy loop as ONE launch, obs_info columns in
    registers / LDS, one grid-wide exchange per added point) against the launch-per-step loop (como_greedy_loop_ws_f32, itself pinned
    to the reference's picks by tests/test_gpu_hotpath.py): the same indices in the same order -- seeded by the area rule (m = 1),
    by three current points, and with early termination (the trace of the largest remaining standard deviation decides where the
    sequence is cut: the same cut)."""
    from como_amd.depth_cov.core import samplers
    g = torch.Generator().manual_seed(11)
    x = torch.rand((1, 1, h, w), generator=g) * 2e-3 + 2e-4
    z = torch.rand((1, 1, h, w), generator=g) * 2e-3 + 2e-4
    o = (torch.rand((1, 1, h, w), generator=g) - 0.5) * 2e-4
    cov = torch.cat((x, o, o, z), dim=1).to(DEV)
    curr = torch.tensor([[[20.0, 30.0], [200.0, 100.5], [120.25, 300.0]]], dtype=torch.float64).to(DEV)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(samplers, "PERSISTENT_LOOP", flag)
        res = []
        for cc, early, thresh in ((None, False, -1e8), (curr, False, -1e8), (curr, True, 0.55), (None, True, 0.6)):
            c, inds = samplers.sample_sparse_coords(cov, 64, "greedy_conditional_entropy", thresh, border=3, dist_thresh=0.05,
                                                    signal_var=1.0, fixed_var=0.0, curr_coords=cc, terminate_early=early)
            res.append(inds.cpu())
        samplers.check_pending_info(wait=True)
        out[flag] = res
    report("persistent_sampler", h=h, w=w, picks=[int(r.shape[1]) for r in out[True]])
    for a, b in zip(out[True], out[False]):
        assert a.shape == b.shape and torch.equal(a, b)
    assert out[True][0].shape[1] == 64 and len(set(out[True][0][0].tolist())) == 64


def test_se3_normalisation_kernel_vs_svd():
    """como_se3_normalize_f64 (normalizeSE3_inplace, como/geometry/lie_algebra.py:98-101) against U V^T of the SVD for rotation blocks
    with float32-sized and larger defects: equal to rounding; the result is orthonormal."""
    from como_amd import _lib, synth
    g = torch.Generator().manual_seed(2)
    T = synth.se3_exp(0.7 * torch.randn((33, 6), generator=g, dtype=torch.float64))
    T[:, :3, :3] += 1e-7 * torch.randn((33, 3, 3), generator=g, dtype=torch.float64)
    T[20:, :3, :3] += 1e-3 * torch.randn((13, 3, 3), generator=g, dtype=torch.float64)
    U, _, Vh = torch.linalg.svd(T[:, :3, :3])
    want = U @ Vh
    Td = T.to(DEV).contiguous()
    assert _lib.lib().como_se3_normalize_f64(Td.data_ptr(), Td.shape[0], _lib.stream_ptr(torch.device(DEV))) == 0
    got = Td.cpu()
    e = (got[:, :3, :3] - want).abs().max().item()
    ortho = (got[:, :3, :3] @ got[:, :3, :3].mT - torch.eye(3, dtype=torch.float64)).abs().max().item()
    report("se3_normalize", err_vs_svd=e, orthonormality=ortho)
    assert e < 5e-15 and ortho < 5e-15 and torch.equal(got[:, :3, 3], T[:, :3, 3]) and torch.equal(got[:, 3], T[:, 3])


def test_grouped_assembly_equals_the_per_pair_form(monkeypatch):
    """ba_reduce_assemble_grouped_kernel (csrc/ba.hip: the fixed-point parts of what the pairs of one reference keyframe add to the
    same system entries summed in registers, one pair of integer atomics per entry and group) against the per-pair assembly on a
    window with one-way frames (3-4 pairs per reference keyframe): the SAME bits (integer addition is associative)."""
    import como_amd.odom.backend.photo as photo
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    B, H, W, m = 3, 96, 128, 16
    st = synth.make_window(B=B, H=H, W=W, m=m, dtype=torch.float64, device=DEV, seed=7,
                           predictor=lambda cov, cm: prep_predictor(cov, cm, 1.0))
    st.update(synth.make_recent([0.3, 1.3, 1.6, 0.6, 1.1, 1.45, 1.8, 1.2], H, W, 7, device=DEV))    # (a group of > 4 pairs: the
    cfg = copy.deepcopy(DEFAULT_CFG)                                                                 #  waves of a workgroup wrap)
    cfg["photo_construction"]["nonmax_suppression_window"] = 2
    monkeypatch.setattr(photo, "ASM_GROUPED_MIN_PAIRS", 0)
    out = {}
    for flag in (True, False, True):
        monkeypatch.setattr(photo, "ASM_GROUPED", flag)
        wb = WindowBA(st, cfg=cfg, pix_dtype=torch.float64, window_full=True)
        Hm, g = wb.linearize()
        torch.cuda.synchronize()
        out.setdefault(flag, []).append((Hm.clone(), g.clone(), wb.table.n_asm_grp, wb.table.b))
    (H1, g1, ng, b), (H1b, g1b, _, _) = out[True]
    (H0, g0, _, _), = out[False]
    assert ng == B and b > 2 * (B - 1)                        # one-way pairs joined the keyframes' groups
    assert torch.equal(H1, H1b) and torch.equal(g1, g1b)
    report("grouped_assembly", pairs=b, groups=ng, H_max_abs_diff=(H1 - H0).abs().max().item(), g_max_abs_diff=(g1 - g0).abs().max().item())
    assert torch.equal(H1, H0) and torch.equal(g1, g0)


@pytest.mark.parametrize("pix,win", [(torch.float64, 1), (torch.float64, 2), (torch.float32, 1), (torch.float32, 4)])
def test_residual_pass_fused_into_the_dense_reference(pix, win, monkeypatch):
    """como_dense_ref_fused_* (csrc/densify.hip DRFuse: pass 1 of batch_photo_cost -- warp, sample, residual, validity, first digit
    of the robust scale, photo.py:104-128 -- inside the dense-reference launch) against the separate residual kernel: residuals,
    validity bytes, the robust scale and the normal equations are IDENTICAL, bit for bit; with one-way frames (pairs whose target
    is not a keyframe) and without."""
    import como_amd.odom.backend.photo as photo
    import como_amd.odom.window_ba as wba
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    B, H, W, m = 3, 96, 128, 16
    st = synth.make_window(B=B, H=H, W=W, m=m, dtype=torch.float64, device=DEV, seed=9,
                           predictor=lambda cov, cm: prep_predictor(cov, cm, 1.0))
    if win == 2:
        st.update(synth.make_recent([0.4, 1.5], H, W, 9, device=DEV))
    cfg = copy.deepcopy(wba.DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = win
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(wba, "_FUSE_PASS1", flag)
        wb = wba.WindowBA(st, cfg=cfg, pix_dtype=pix, window_full=True)
        Hm, g = wb.linearize()
        torch.cuda.synchronize()
        aux = photo.last_aux
        out[flag] = (aux["r"].clone(), aux["valid"].clone(), wb.sigma.clone(), Hm.clone(), g.clone(), wb.table.b)
        d = wb.iterate()                                   # (a second pass through the cached argument block)
        out[flag] += (d.clone(),)
    a, b_ = out[True], out[False]
    v = b_[1].bool()
    report("fused_pass1", pix=str(pix), window=win, pairs=a[5], valid=int(v.sum()), sigma=float(a[2][0]))
    assert torch.equal(a[1], b_[1]) and int(v.sum()) > 1000
    assert torch.equal(a[0][v], b_[0][v]) and torch.equal(a[0], b_[0])
    assert torch.equal(a[2], b_[2]) and torch.equal(a[3], b_[3]) and torch.equal(a[4], b_[4]) and torch.equal(a[6], b_[6])


def test_network_on_the_side_stream_equals_the_inline_call():
    """Mapping.start_model / take_model (the covariance network of a keyframe insertion submitted on a side stream as soon as the
    tracker asks for a keyframe, joined where add_keyframe first reads it) against run_model in line: the same covariance image,
    bit for bit, with main-stream work queued in between; a pending result of another frame size is not handed out."""
    import types
    from como_amd import synth
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule, run_model
    from como_amd.odom.Mapping import Mapping
    model = DepthCovModule({k: v.to(DEV) for k, v in synth.depthcov_state_dict(3).items()})
    g = torch.Generator().manual_seed(5)
    rgb = torch.rand((1, 3, 120, 160), generator=g).to(DEV)
    ns = types.SimpleNamespace(run_model=lambda x: run_model(model, x, network_size=[96, 128], dtype=torch.float64))
    want = ns.run_model(rgb).clone()
    for _ in range(2):
        Mapping.start_model(ns, rgb)
        busy = torch.randn(2048, 2048, device=DEV)
        for _ in range(4):
            busy = busy @ busy.mT * 1e-3                   # (main-stream work between the submission and the join)
        got = Mapping.take_model(ns, rgb.double())
        assert ns._net_pending is None and got.dtype == torch.float64 and torch.equal(got, want)
    Mapping.start_model(ns, rgb)
    other = torch.rand((1, 3, 96, 128), generator=g).to(DEV)
    assert torch.equal(Mapping.take_model(ns, other), ns.run_model(other)) and ns._net_pending is None
    torch.cuda.synchronize()
    report("async_network", max_abs=float(want.abs().max()))


def test_frame_record_kernel_equals_the_torch_chain():
    """como_track_frame_record_f32 (the record the host reads back of a tracked frame: world pose, |t|, reprojection statistics,
    level status words in one launch) against the chain it replaces in Tracking._frame_body -- composeSE3 mode 2, linalg.norm, two
    casts, cat: poses and statistics bit for bit, |t| within one float32 ulp of torch's reduction (reported; it only feeds the
    keyframe thresholds)."""
    from como_amd import _lib
    from como_amd.geometry.lie_algebra import composeSE3
    from como_amd.synth import se3_exp
    g = torch.Generator().manual_seed(11)
    worst = 0.0
    for trial in range(20):
        T = se3_exp(0.3 * torch.randn((1, 6), generator=g, dtype=torch.float64)).float().to(DEV)
        Tw = se3_exp(0.5 * torch.randn((1, 6), generator=g, dtype=torch.float64)).float().to(DEV)
        aff = (0.1 * torch.randn((1, 2, 1), generator=g)).to(DEV)
        nl = 3
        recs = torch.randn((nl, 106), generator=g).to(DEV)
        recs[:, 104] = torch.tensor([0.0, -1.0, -2.0])[:nl]
        med3 = torch.rand((1, 3), generator=g).to(DEV)
        nseen = torch.tensor([123456 + trial], dtype=torch.int32, device=DEV)
        sc = torch.empty(3 + nl + 34, device=DEV)
        _lib.check(_lib.lib().como_track_frame_record_f32(T.data_ptr(), aff.data_ptr(), Tw.data_ptr(), med3.data_ptr(), nseen.data_ptr(),
                                                          recs.data_ptr(), nl, 106, sc.data_ptr(), _lib.stream_ptr(torch.device(DEV))), "record")
        T_w = composeSE3(Tw, T, 2)
        want = torch.cat((torch.linalg.norm(T[:, :3, 3]).reshape(1), med3[0, 0].reshape(1), nseen.float(), recs[:, 104], T.reshape(-1),
                          aff.reshape(-1), T_w.reshape(-1)))
        assert torch.equal(sc[1:], want[1:])
        ulp = abs(float(sc[0]) - float(want[0])) / float(torch.finfo(torch.float32).eps * want[0].abs())
        worst = max(worst, ulp)
    report("frame_record", norm_worst_ulp=worst)
    assert worst <= 1.0


def test_reference_pyramid_in_one_launch_equals_the_per_level_chain():
    """como_track_reference_pyr_f32 (pose composition + nearest-neighbour depth pyramid + the reference arrays of every level in one
    launch) against the chain it replaces in Tracking.update_kf_reference -- composeSE3 mode 1, pyr_depth, one
    como_track_reference_f32 per level: points, masks, Jacobians bit for bit; two reference keyframes, odd image sizes."""
    import ctypes
    from como_amd import _lib
    from como_amd.geometry.lie_algebra import composeSE3, se3_exp
    from como_amd.utils.image_processing import pyr_depth
    g = torch.Generator().manual_seed(3)
    for (H0, W0, nl, nk) in ((48, 64, 3, 1), (51, 67, 3, 2), (30, 40, 2, 2), (24, 32, 4, 1)):
        depth = (0.5 + torch.rand((nk, 1, H0, W0), generator=g)).to(DEV)
        poses = se3_exp(0.2 * torch.randn((nk, 6), generator=g, dtype=torch.float64)).float().to(DEV)
        dpyr, lvl = [], depth
        for i in range(nl - 1):
            dpyr.insert(0, lvl)
            lvl = pyr_depth(lvl, "nearest_neighbor", kernel_size=2)
        dpyr.insert(0, lvl)
        rel = composeSE3(poses[nk - 1:nk], poses, 1).contiguous()
        Ks, grads, vals, want, got = [], [], [], [], []
        for i, d in enumerate(dpyr):
            h, w = d.shape[-2:]
            sc = 2.0 ** (nl - 1 - i)
            K = torch.tensor([[500.0 / sc, 0, W0 / 2 / sc], [0, 480.0 / sc, H0 / 2 / sc], [0, 0, 1]], device=DEV)
            gr = torch.randn((nk, h * w, 1, 2), generator=g).to(DEV)
            va = torch.rand((nk, h * w, 1), generator=g).to(DEV)
            P = torch.empty((nk, h * w, 3), device=DEV)
            M = torch.empty((nk, h * w), dtype=torch.uint8, device=DEV)
            J = torch.empty((nk, h * w, 1, 8), device=DEV)
            _lib.check(_lib.lib().como_track_reference_f32(d.contiguous().data_ptr(), rel.data_ptr(), K.data_ptr(), gr.data_ptr(), va.data_ptr(),
                                                           nk, h, w, 50.0, 1e-4, P.data_ptr(), M.data_ptr(), J.data_ptr(),
                                                           _lib.stream_ptr(torch.device(DEV))), "ref")
            Ks.append(K); grads.append(gr); vals.append(va); want.append((P, M, J))
            got.append((torch.empty_like(P), torch.empty_like(M), torch.empty_like(J)))
        arr = lambda ts: (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ts])
        hw = (ctypes.c_int * (2 * nl))(*[x for d in dpyr for x in d.shape[-2:]])
        _lib.check(_lib.lib().como_track_reference_pyr_f32(depth.data_ptr(), H0, W0, poses.contiguous().data_ptr(), nk, nl, hw, arr(Ks), arr(grads),
                                                           arr(vals), arr([t[0] for t in got]), arr([t[1] for t in got]), arr([t[2] for t in got]),
                                                           50.0, 1e-4, _lib.stream_ptr(torch.device(DEV))), "pyr")
        torch.cuda.synchronize()
        for (P, M, J), (P2, M2, J2) in zip(want, got):
            assert torch.equal(P, P2) and torch.equal(M, M2) and torch.equal(J, J2)
        assert int(sum(int(t[1].sum()) for t in got)) > 0
    # a size that is not the finest size pooled: refused
    bad = (ctypes.c_int * (2 * nl))(*([1, 1] * nl))
    assert _lib.lib().como_track_reference_pyr_f32(depth.data_ptr(), H0, W0, poses.data_ptr(), nk, nl, bad, arr(Ks), arr(grads), arr(vals),
                                                   arr([t[0] for t in got]), arr([t[1] for t in got]), arr([t[2] for t in got]), 50.0, 1e-4,
                                                   _lib.stream_ptr(torch.device(DEV))) != 0
    report("reference_pyramid", cases=4)


# ---------------------------------------------------------------------------------------------------------------------
def test_median_streamed_ahead_equals_the_in_iteration_one(monkeypatch):
    """`WindowBA._issue_median_ahead` (csrc/window.hip como_win_logz_ahead + the depth-only pass, on the side stream right after an
    iteration's update) against the full-image median computed inside the next iteration (Mapping.store_vars, Mapping.py:749-758):
    (i) three iterations of one window, then a re-targeted set of one-way frames, with and without the switch: state, normal
    equations and medians identical bit for bit; (ii) the 72-frame sequence of the ATE fixture in check mode (every adopted median
    is also computed the usual way and compared): same decisions as the reference's loop, no mismatch."""
    import como_amd.odom.window_ba as wba
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from tests.conftest import load_golden
    B, H, W, m = 3, 96, 128, 16
    st = synth.make_window(B=B, H=H, W=W, m=m, dtype=torch.float64, device=DEV, seed=11,
                           predictor=lambda cov, cm: prep_predictor(cov, cm, 1.0))
    st.update(synth.make_recent([0.3, 1.3], H, W, 11, device=DEV))
    cfg = copy.deepcopy(wba.DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 2
    res = []
    for ahead in (False, True):
        for k in wba.AHEAD_STATS:
            wba.AHEAD_STATS[k] = 0
        wb = wba.WindowBA({k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}, cfg=cfg, pix_dtype=torch.float64,
                          window_full=True, rec_capacity=4, band_median=False)
        assert wb.full_median
        wb.median_ahead = ahead
        for _ in range(3):
            wb.iterate()
        s2 = dict(st)
        s2.update(synth.make_recent([0.3, 1.3, 1.6], H, W, 11, device=DEV))
        sn = wb.snapshot_state()
        s2["kf_poses"], s2["kf_aff_params"] = sn["poses"][:B].clone(), sn["aff"][:B].reshape(B, 2, 1).clone()
        s2["P_m"], s2["median_depth_init"] = sn["P_m"].clone(), sn["median"].clone()
        s2["recent_poses"] = torch.cat((sn["poses"][B:B + 2], s2["recent_poses"][2:]))
        s2["kf_img_and_grads"], s2["correspondence_mask"] = wb._src_kf_img, wb._src_mask
        s2["_published_by"] = wb                             # (what Mapping.iterate records: the keyframe rows are this buffer's own)
        assert wb.retarget(s2) is True
        wb.iterate()
        torch.cuda.synchronize()
        assert int(wb.info.item()) == 0
        # (med3_full itself is not compared: with the switch on it already holds the NEXT iteration's median)
        res.append((wb.state_flat.clone(), wb.H.clone(), wb.g.clone(), dict(wba.AHEAD_STATS)))
    off, on = res
    report("median_ahead_window", stats=on[3], state_diff=(off[0] - on[0]).abs().max().item())
    assert off[3]["issued"] == 0 and on[3]["issued"] == 4 and on[3]["adopted"] == 3
    assert all(torch.equal(a, b) for a, b in zip(off[:3], on[:3]))
    # a state written from outside drops the streamed median
    s3 = dict(s2)
    s3["_published_by"] = None
    assert wb.retarget(s3) is True and wb._ahead is None
    # --- the sequential loop, check mode
    import scripts.ate_sequence as ats
    from scripts.ate_sequence import run_ate_sequence
    loop_cfgs = ats.loop_cfgs
    monkeypatch.setattr(wba, "_MEDIAN_AHEAD", 2)
    G = load_golden("ate_sequence.npz")
    for mode, least in (("gap", 8), ("end", 30)):           # "gap": one-way frames only (the sequential loop's default)
        def cfgs_with_ahead(*a, **k):
            c = loop_cfgs(*a, **k)
            c["mapping"]["median_ahead"] = mode
            return c
        monkeypatch.setattr(ats, "loop_cfgs", cfgs_with_ahead)
        for k in wba.AHEAD_STATS:
            wba.AHEAD_STATS[k] = 0
        kinds, poses, odo = run_ate_sequence(G, "float")
        stats = dict(wba.AHEAD_STATS)
        report("median_ahead_loop", mode=mode, **stats)
        assert kinds == [int(x) for x in G["kinds"]]
        assert stats["adopted"] >= least and stats["checked"] == stats["adopted"] and stats["mismatch"] == 0


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,W,masked", [(480, 640, False), (240, 320, True), (120, 160, False), (60, 80, True), (37, 53, False)])
def test_band_split_level_kernel_vs_exact_form(H, W, masked):
    """csrc/track.hip: the band-split sums of the persistent level kernel (inlier / outlier sums formed while the last digit of the
    median is still open, the undecided pixels listed and added exactly afterwards: two device-wide synchronisations per
    iteration) against the exact form of the same kernel (`como_track_level_set_split(0)`: scale first, sums afterwards) and
    against the per-iteration chain: same iteration count and stop decision, pose / affine within float32 summation noise --
    also when the list overflows (`como_track_level_debug_amb_cap`: 0 and 1 entries -> the exact form runs after the split pass)."""
    import como_amd.odom.frontend.photo_tracking as pt
    from como_amd import _lib
    from tests.test_gpu_r2 import _tracking_level_inputs
    L = _lib.lib()
    L.como_track_level_workspace_bytes()                       # (reads COMO_TRACK_SPLIT on the first launch only: prime the switch below)
    tp, K, P, vals, J = _tracking_level_inputs(H, W, 5)
    aff = torch.zeros((1, 2, 1), device=DEV)
    mask = None
    if masked:
        g = torch.Generator().manual_seed(2)
        mask = (torch.rand(P.shape[1], generator=g) < 0.7).to(torch.uint8).to(DEV)
    term = {"max_iter": 50, "delta_norm": 1e-3, "rel_tol": 1e-3, "grad_norm": 1.0}
    term6 = {"max_iter": 6, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}

    def run(tc, fused):
        T, a = pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J.clone(), tp["img_cur"], K, 0.1, tc, in_mask=mask, fused=fused)
        if not fused:
            return T, a, pt.photo_level_tracking.last_iters, 0
        rec = pt.photo_level_tracking.last_out.cpu()
        return T, a, int(rec[105]), int(rec[104])

    Tc, ac, itc, _ = run(term, False)
    T6c, _, _, _ = run(term6, False)
    out = {}
    prev = L.como_track_level_set_split(0)
    try:
        for name, split, cap in (("split", 1, -1), ("exact", 0, -1), ("overflow0", 1, 0), ("overflow1", 1, 1)):
            L.como_track_level_set_split(split)
            L.como_track_level_debug_amb_cap(cap)
            out[name] = run(term, True), run(term6, True)
    finally:
        L.como_track_level_set_split(prev)
        L.como_track_level_debug_amb_cap(-1)
    errs = {n: ((r[0][0] - Tc).abs().max().item(), (r[0][1] - ac).abs().max().item(), (r[1][0] - T6c).abs().max().item())
            for n, r in out.items()}
    report("band_split_level", H=H, W=W, masked=masked, iters_chain=itc, iters={n: r[0][2] for n, r in out.items()},
           status={n: r[0][3] for n, r in out.items()}, errs=errs,
           split_vs_exact=(out["split"][0][0] - out["exact"][0][0]).abs().max().item())
    for n, r in out.items():
        assert r[0][3] == 0 and r[1][3] == 0 and r[0][2] == itc and r[1][2] == 6, n
        assert max(errs[n]) < 2e-6, n


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,W", [(480, 640), (97, 131), (192, 256)])
def test_frame_pyramid_in_one_launch_equals_the_chain(H, W):
    """csrc/image.hip frame_pyramid3_kernel (the head of the tracker's frame graph: luma + both blur_down levels + the clears of the
    level kernels' barrier workspaces, one launch) against Tracking.prep_tracking_img's chain (rgb_to_grayscale + ImagePyramidModule:
    three launches): the three images are identical bit for bit, odd sizes included; the listed buffers are cleared."""
    import ctypes
    from como_amd import _lib
    from como_amd.utils import image_processing as ip
    g = torch.Generator().manual_seed(H + W)
    rgb = torch.rand((1, 3, H, W), generator=g).to(DEV)
    ref = ip.ImagePyramidModule(1, 0, 3, DEV, torch.float32)(ip.rgb_to_grayscale(rgb))            # [coarse .. fine]
    H1, W1 = (H + 1) // 2, (W + 1) // 2
    gray = torch.empty((1, 1, H, W), device=DEV)
    l1 = torch.empty((1, 1, H1, W1), device=DEV)
    l2 = torch.empty((1, 1, (H1 + 1) // 2, (W1 + 1) // 2), device=DEV)
    bufs = [torch.full((n,), 7, dtype=torch.int32, device=DEV) for n in (4, 26432, 4096 * 3)]
    ptrs = (ctypes.c_void_p * 3)(*[b.data_ptr() for b in bufs])
    nb = (ctypes.c_long * 3)(*[b.numel() * 4 for b in bufs])
    rc = _lib.lib().como_track_frame_pyramid3_f32(rgb.data_ptr(), gray.data_ptr(), l1.data_ptr(), l2.data_ptr(), H, W, ptrs, nb, 3,
                                                  _lib.stream_ptr(torch.device(DEV)))
    assert rc == 0
    torch.cuda.synchronize()
    assert [tuple(t.shape) for t in ref] == [tuple(l2.shape), tuple(l1.shape), tuple(gray.shape)]
    assert torch.equal(ref[2], gray) and torch.equal(ref[1], l1) and torch.equal(ref[0], l2)
    assert all(int(b.abs().max()) == 0 for b in bufs)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,W,c,masked", [(60, 80, 1, False), (60, 80, 1, True), (37, 53, 1, False), (30, 40, 3, True), (24, 32, 1, False)])
def test_one_workgroup_level_kernel(H, W, c, masked):
    """csrc/track.hip track_level_one_kernel: a level of at most 4800 elements (80x60 gray, the coarsest level of a 640x480 frame)
    in ONE workgroup -- pixel waves + one solver wave, LDS histograms, workgroup barriers only -- in its three workgroup sizes against
    the multi-workgroup form of the level kernel (`como_track_level_set_one(0)`) and the per-iteration chain: same iteration
    count and stop decision, pose / affine within float32 summation noise, the same bits on a second launch; one more element
    than the capacity goes through the multi-workgroup form."""
    import como_amd.odom.frontend.photo_tracking as pt
    from como_amd import _lib
    from tests.test_gpu_r2 import _tracking_level_inputs
    L = _lib.lib()
    tp, K, P, vals, J = _tracking_level_inputs(H, W, 7)
    img = tp["img_cur"]
    if c > 1:                                             # c channels: one residual per (pixel, channel)
        sc = torch.tensor([1.0, 0.8, 1.1], device=DEV)[:c]
        img = (img[:, :1] * sc.view(1, c, 1, 1)).contiguous()
        vals = (vals * sc.view(1, 1, c)).contiguous()
        Jc = (J.reshape(1, -1, 1, 8) * sc.view(1, 1, c, 1)).contiguous()
        Jc[..., 7] = J.reshape(1, -1, 1, 8)[..., 7]      # (the offset's derivative does not scale with the channel)
        J = Jc
    aff = torch.zeros((1, 2, 1), device=DEV)
    mask = None
    if masked:
        g = torch.Generator().manual_seed(4)
        mask = (torch.rand(P.shape[1], generator=g) < 0.7).to(torch.uint8).to(DEV)
    term = {"max_iter": 50, "delta_norm": 1e-3, "rel_tol": 1e-3, "grad_norm": 1.0}
    term6 = {"max_iter": 6, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}

    def run(tc, fused):
        T, a = pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J.clone(), img, K, 0.1, tc, in_mask=mask, fused=fused)
        if not fused:
            return T.clone(), a.clone(), pt.photo_level_tracking.last_iters, 0
        rec = pt.photo_level_tracking.last_out.cpu()
        return T.clone(), a.clone(), int(rec[105]), int(rec[104])

    Tc, ac, itc, _ = run(term, False)
    T6c, _, _, _ = run(term6, False)
    out = {}
    prev = L.como_track_level_set_one(0)
    try:
        for name, nt in (("multi", 0), ("one1024", 1024), ("one768", 768), ("one512", 512)):
            L.como_track_level_set_one(nt)
            out[name] = run(term, True), run(term6, True), run(term, True)
    finally:
        L.como_track_level_set_one(prev)
    errs = {n: ((r[0][0] - Tc).abs().max().item(), (r[0][1] - ac).abs().max().item(), (r[1][0] - T6c).abs().max().item())
            for n, r in out.items()}
    report("one_workgroup_level", H=H, W=W, c=c, masked=masked, iters_chain=itc, iters={n: r[0][2] for n, r in out.items()},
           status={n: r[0][3] for n, r in out.items()}, errs=errs,
           one_vs_multi=(out["one1024"][0][0] - out["multi"][0][0]).abs().max().item())
    for n, r in out.items():
        assert r[0][3] == 0 and r[1][3] == 0 and r[0][2] == itc and r[1][2] == 6, n
        assert max(errs[n]) < 2e-6, n
        assert torch.equal(r[0][0], r[2][0]) and torch.equal(r[0][1], r[2][1]), n        # repeatable bits


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,cin,cout,H,W", [(1, 3, 16, 192, 256), (1, 32, 16, 192, 256), (2, 64, 32, 96, 128), (1, 64, 32, 48, 64), (2, 16, 32, 20, 28)])
def test_groupnorm_finalised_inside_the_convolution(N, cin, cout, H, W):
    """csrc/nn.hip gn_arrive_finalize: the scale / shift of the GroupNorm behind a wide-level convolution formed by that
    convolution's LAST wave (returning agent-scope adds, an arrival counter behind the statistics) against the separate
    `como_nn_gn_finalize_f32` launch -- the same scsh up to the arrival order of the float64 adds, on the three tile shapes of the
    LDS-tiled kernel and on the generic kernel; repeated, because a finalisation that ran before every add had landed would show
    as an occasional large error, not a constant one."""
    from como_amd.depth_cov.nn import UNet as U
    from como_amd import _lib
    from tests.conftest import rel_err
    g = torch.Generator().manual_seed(cin * 100 + cout)
    conv = U._Conv(torch.randn(cout, cin, 3, 3, generator=g).to(DEV) / (cin * 9) ** 0.5, (torch.randn(cout, generator=g) * 0.1).to(DEV))
    gamma = (1.0 + 0.2 * torch.randn(cout, generator=g)).to(DEV).contiguous()
    beta = (0.2 * torch.randn(cout, generator=g)).to(DEV).contiguous()
    x = torch.randn(N, cin, H, W, generator=g).to(DEV)
    prev = U.GN_IN_CONV
    worst = 0.0
    try:
        U.GN_IN_CONV = False
        y0, sc0 = U._conv3_any(conv, x, norm=(gamma, beta))
        U.GN_IN_CONV = True
        for _ in range(25):
            y1, sc1 = U._conv3_any(conv, x, norm=(gamma, beta))
            worst = max(worst, rel_err(sc1, sc0))
            assert torch.equal(y1, y0)
    finally:
        U.GN_IN_CONV = prev
    # against torch's statistics of the same output
    m = y0.double().reshape(N, 16, -1).mean(-1)
    v = y0.double().reshape(N, 16, -1).var(-1, unbiased=False)
    rstd = (1.0 / torch.sqrt(v + 1e-5)).repeat_interleave(cout // 16, 1)
    sc_ref = rstd * gamma.double()[None]
    sh_ref = beta.double()[None] - m.repeat_interleave(cout // 16, 1) * sc_ref
    ref = torch.stack((sc_ref, sh_ref), -1).float()
    report("gn_in_conv", N=N, cin=cin, cout=cout, H=H, W=W, vs_separate_launch=worst, vs_torch=rel_err(sc1, ref))
    assert worst < 1e-6 and rel_err(sc1, ref) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
def test_keyframe_glue_kernels_vs_torch_chains():
    """csrc/kfglue.hip: the element-wise chains of a keyframe insertion as one launch each -- shift + 1 / sqrt of the conditional
    variances (min over the rows that count), the observations' validity as zero weights, the correspondence test, the coordinate
    normalisation -- against the torch op chains they replace: the SAME bits (one rounding per op, same order), masked rows and
    non-finite / non-positive depths included."""
    import como_amd.depth_cov.core.distill_depth as dd
    from como_amd import _lib
    from como_amd.utils import coords as cu
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    n = 70001
    # ---- predictor_sinv
    var = (torch.rand(1, n, generator=g, dtype=torch.float64) - 0.01).to(DEV)
    mask = (torch.rand(n, generator=g) < 0.6).to(DEV)
    for rm in (None, mask):
        vmin = torch.min(var) if rm is None else torch.min(torch.where(rm.reshape(1, n), var, torch.full_like(var, float("inf"))))
        ref = 1.0 / torch.sqrt((var + (vmin + 1e-8)).unsqueeze(-1))
        out = torch.empty((1, n, 1), dtype=torch.float64, device=DEV)
        part = torch.empty(64, dtype=torch.float64, device=DEV)
        _lib.check(L.como_kf_predictor_sinv_f64(var.data_ptr(), _lib.ptr(rm), n, part.data_ptr(), out.data_ptr(), _lib.stream_ptr(var.device)), "sinv")
        assert torch.equal(out, ref) or bool(((out == ref) | (out.isnan() & ref.isnan())).all())
    # ---- distill_prep
    z = (torch.rand(1, n, 1, generator=g, dtype=torch.float64) * 3.0 - 0.3).to(DEV)
    z[0, 5, 0] = float("nan"); z[0, 6, 0] = float("inf"); z[0, 7, 0] = 0.0
    sinv = (0.5 + torch.rand(1, n, 1, generator=g, dtype=torch.float64)).to(DEV)
    for om in (None, mask):
        for weighted, s_t, s_sc in ((True, sinv, None), (True, None, 0.07), (True, None, torch.tensor(0.07, dtype=torch.float64, device=DEV)),
                                    (False, None, None)):
            okm = z[:, :, 0:1] > 0.1
            if om is not None:
                okm = okm & om.reshape(1, -1, 1)
            zs = torch.where(okm, z, torch.ones_like(z))
            y = torch.log(zs)
            sv = s_t if s_t is not None else ((1.0 / s_sc) * torch.ones_like(z) if s_sc is not None else torch.ones_like(z))
            w = torch.where(okm, sv * sv, torch.zeros_like(sv)) if weighted else okm.to(torch.float64)
            ok2, zs2, y2, w2 = dd.distill_prep(z, om, 0.1, s_t, s_sc, weighted, want_zs=True)
            assert torch.equal(ok2, okm) and torch.equal(zs2, zs) and torch.equal(y2, y) and torch.equal(w2, w)
            # the depth column of (1,n,3) points: a strided view
            P3 = torch.zeros((1, n, 3), dtype=torch.float64, device=DEV)
            P3[:, :, 2:3] = z
            ok3, zs3, y3, w3 = dd.distill_prep(P3[:, :, 2:3], om, 0.1, s_t, s_sc, weighted, want_zs=True)
            assert torch.equal(ok3, okm) and torch.equal(zs3, zs) and torch.equal(y3, y) and torch.equal(w3, w)
    # ---- corr_good
    m = 257
    P = [(0.2 + 2.0 * torch.rand(1, m, 3, generator=g, dtype=torch.float64)).to(DEV) for _ in range(4)]
    P[1][0, 3, 2] = -1.0                                   # log of a negative depth: NaN -> not good
    P[2][0, 4, 2] = 0.0
    for k in range(0, m, 2):
        P[1][0, k, 2] = P[0][0, k, 2] * 1.01
        P[3][0, k, 2] = P[2][0, k, 2] * 0.995
    grad = torch.rand(1, m, 1, generator=g, dtype=torch.float64).to(DEV)
    err = torch.maximum(torch.abs(torch.log(P[0][..., 2:3]) - torch.log(P[1][..., 2:3])), torch.abs(torch.log(P[2][..., 2:3]) - torch.log(P[3][..., 2:3])))
    ref = ((err < 0.02) & (grad < 0.5))[0, :, 0]
    good = torch.empty((m,), dtype=torch.bool, device=DEV)
    _lib.check(L.como_kf_corr_good_f64(P[0].data_ptr() + 16, P[1].data_ptr() + 16, P[2].data_ptr() + 16, P[3].data_ptr() + 16, 3,
                                       grad.reshape(-1).data_ptr(), m, 0.02, 0.5, good.data_ptr(), _lib.stream_ptr(grad.device)), "corr_good")
    assert torch.equal(good, ref) and 0 < int(ref.sum()) < m
    # ---- normalize_coords (both element types; the fused path is what normalize_coordinates takes for contiguous (.., 2) tensors)
    for dt in (torch.float32, torch.float64):
        x = (torch.rand(1, 4099, 2, generator=g, dtype=torch.float64) * 600.0).to(dt).to(DEV)
        A, A2 = cu._inv_dims((480, 640), x.device, dt)
        ref = A2 * x + A - 1
        prev = cu._KF_GLUE
        try:
            cu._KF_GLUE = True
            got = cu.normalize_coordinates(x, (480, 640))
        finally:
            cu._KF_GLUE = prev
        assert torch.equal(got, ref)
        got_s = cu.normalize_coordinates(x, (480, 640), swap=True)
        assert torch.equal(got_s, cu.swap_coords_xy(ref))
        # ---- gradient magnitude, affine brightness composition (both element types)
        from como_amd.odom.frontend import corr as cr
        from como_amd.geometry import affine_brightness as ab
        gx = torch.randn(1, 1, 37, 53, generator=g, dtype=torch.float64).to(dt).to(DEV)
        gy = torch.randn(1, 1, 37, 53, generator=g, dtype=torch.float64).to(dt).to(DEV)
        assert torch.equal(cr._grad_mag(gx, gy), torch.sqrt(gx * gx + gy * gy))
        p_ = (0.3 * torch.randn(3, 2, 1, generator=g, dtype=torch.float64)).to(dt).to(DEV)
        q_ = (0.3 * torch.randn(3, 2, 1, generator=g, dtype=torch.float64)).to(dt).to(DEV)
        prev = ab._KF_GLUE
        try:
            ab._KF_GLUE = False
            r0, r1 = ab.get_aff_w_curr(p_, q_), ab.get_rel_aff(p_, q_)
            ab._KF_GLUE = True
            f0, f1 = ab.get_aff_w_curr(p_, q_), ab.get_rel_aff(p_, q_)
        finally:
            ab._KF_GLUE = prev
        assert torch.equal(f0, r0) and torch.equal(f1, r1)
    # ---- the small system of the conditional distillation
    m1, m2, mp = 23, 37, 60
    z1 = (0.3 + torch.rand(1, m1, 1, generator=g, dtype=torch.float64)).to(DEV)
    AtA = torch.randn(1, mp, mp, generator=g, dtype=torch.float64).to(DEV)
    AtA[0, m1 + 3, m1 + 5] = -0.0
    Atb = torch.randn(1, mp, 1, generator=g, dtype=torch.float64).to(DEV)
    s_med = torch.log(torch.tensor(1.7, dtype=torch.float64, device=DEV))
    sp2 = (1.0 / 5e-2) ** 2
    c_ref = torch.cat((torch.log(z1).reshape(1, m1, 1), torch.zeros((1, mp - m1, 1), device=DEV, dtype=torch.float64)), dim=1)
    A_ref = AtA[:, m1:m1 + m2, m1:m1 + m2] + sp2 * torch.eye(m2, device=DEV, dtype=torch.float64)
    b_ref = Atb[:, m1:m1 + m2] + sp2 * s_med
    c_ = torch.empty((1, mp, 1), dtype=torch.float64, device=DEV)
    A_ = torch.empty((1, m2, m2), dtype=torch.float64, device=DEV)
    b_ = torch.empty((1, m2, 1), dtype=torch.float64, device=DEV)
    st = _lib.stream_ptr(c_.device)
    _lib.check(L.como_kf_cond_c_f64(z1.data_ptr(), m1, mp, c_.data_ptr(), st), "cond_c")
    _lib.check(L.como_kf_cond_system_f64(AtA.data_ptr(), Atb.data_ptr(), mp, m1, m2, sp2, s_med.data_ptr(), A_.data_ptr(), b_.data_ptr(), st), "cond_system")
    same_bits = lambda a, b: torch.equal(a.view(torch.int64), b.contiguous().view(torch.int64))
    assert torch.equal(c_, c_ref) and same_bits(A_, A_ref) and same_bits(b_, b_ref)
    # ---- masked standard deviation (a reduction: fixed order of its own, equal to torch.std of the gathered entries to rounding)
    resid = torch.randn(1, n, 1, generator=g, dtype=torch.float64).to(DEV) * 0.03 + 0.2
    okb = (torch.rand(1, n, 1, generator=g) < 0.7).to(DEV)
    resid[0, 9, 0] = float("nan")
    okb[0, 9, 0] = False                                   # a masked row may hold anything
    mr = dd.MaskedResidual(resid, okb)
    ref_std = torch.std(resid[okb])
    got_std = mr.std()
    assert got_std.shape == () and abs(got_std.item() - ref_std.item()) < 1e-13 * ref_std.item()
    assert torch.equal(got_std, mr.std())                  # the same bits every time
    report("kf_glue", sinv="equal", distill_prep="equal", corr_good="equal", normalize="equal", grad_mag="equal", aff="equal", cond="equal",
           masked_std_rel=abs(got_std.item() - ref_std.item()) / ref_std.item())
