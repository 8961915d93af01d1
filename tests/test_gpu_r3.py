"""Round-3 pins of the HIP path (through the C ABI), `-m gpu` on an MI355X:
  * the compact dense reference (como_ba_args.zmode 2: nine planes per pixel, the pose Jacobian rebuilt inside the block
    kernels) against the materialised forms (zmode 0 = the reference's own argument list) and the reference's golden system;
  * the small-system conditioning kernels (csrc/smallsolve.hip) against torch.linalg;
  * non-finite shares of the sharded exchange poison the system whatever the number of ranks (advisor finding: an additive
    sentinel wrapped to zero at 4 / 8 ranks);
  * the persistent tracking kernel's result record starts as the initial pose."""
import pytest
import torch

from tests.conftest import load_golden, rel_err, report, scaled_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


def _compact_inputs(G):
    """Per-keyframe inputs of the factored path from a ba_window_* fixture (the reference's own intermediate tensors)."""
    B, n = G["coords_n"].shape[:2]
    m = G["coords_m"].shape[1]
    Hh, Ww = G["kf_img_and_grads"].shape[-2:]
    cn = G["coords_n"]
    pixidx = (cn[..., 0] * Ww + cn[..., 1]).to(torch.int32)
    dl_m = (G["dlogzm_dzm"] @ G["dzm_dTwc"]).reshape(B, m, 6)                      # dlogz_m / dT_wc
    return B, n, m, Hh, Ww, pixidx, dl_m


@pytest.mark.parametrize("name,tol", [("ba_window_f64.npz", 1e-12), ("ba_window_f32.npz", 2e-5)])
def test_compact_dense_reference_vs_golden(name, tol):
    """como_dense_ref_* flag 16: P_w bit-identical to the full form, the six planes = K~[n,:] dlogz_m/dT_wc, and the pose Jacobian
    rebuilt from them -- [-[u]x R, R] + u (x) dl with u = P_w - t_wc -- equals the reference's dPwn_dTwc."""
    from como_amd.odom.backend.dense_ref import dense_reference_factored
    G = load_golden(name)
    B, n, m, Hh, Ww, pixidx, dl_m = _compact_inputs(G)
    Kt = dev(G["Knm_Kmminv"].reshape(B, Hh * Ww, m))
    args = (dev(G["logzm"]), dev(G["kf_poses"]), Kt, dev(pixidx), dev(G["intrinsics"][0]), dev(dl_m), Ww)
    Pw_full, dT_full, uvec, med_full, _ = dense_reference_factored(*args)
    Pw_full, dT_full, med_full = Pw_full.clone(), dT_full.clone(), med_full.clone()
    Pw, dl, none, med, _ = dense_reference_factored(*args, compact=True)
    assert none is None and dl.shape == (B, 6, n)
    assert torch.equal(Pw, Pw_full) and torch.equal(med, med_full)                  # same arithmetic for what feeds masks / medians
    rows = G["Knm_Kmminv"].reshape(B, -1, m)[torch.arange(B)[:, None], pixidx.long()]          # (B,n,m)
    dl_ref = torch.einsum("bnm,bmk->bkn", rows.double(), dl_m.double())
    assert rel_err(dl, dl_ref) < tol * 10
    # rebuild dP_w/dT_wc on the host exactly as ref_pose_geom does, compare with the reference's tensor
    R, t = G["kf_poses"][:, :3, :3].double(), G["kf_poses"][:, :3, 3].double()
    u = Pw.cpu().double().permute(0, 2, 1) - t[:, None, :]                          # (B,n,3)
    ux = torch.zeros(B, n, 3, 3, dtype=torch.float64)
    ux[..., 0, 1], ux[..., 0, 2], ux[..., 1, 0] = -u[..., 2], u[..., 1], u[..., 2]
    ux[..., 1, 2], ux[..., 2, 0], ux[..., 2, 1] = -u[..., 0], -u[..., 1], u[..., 0]
    dT = torch.cat((-(ux @ R[:, None]), R[:, None].expand(B, n, 3, 3)), dim=-1) + u[..., None] * dl.cpu().double().permute(0, 2, 1)[:, :, None, :]
    report("dense_ref_compact", case=name, dl=rel_err(dl, dl_ref), dT=rel_err(dT, G["dPwn_dTwc"]))
    assert rel_err(dT, G["dPwn_dTwc"]) < tol * 20


@pytest.mark.parametrize("name,tol", [("ba_window_f64.npz", 1e-10), ("ba_window_f32.npz", 3e-4)])
@pytest.mark.parametrize("variant", [0, 1])
def test_compact_block_kernels_match_reference_signature_path(name, tol, variant, monkeypatch):
    """zmode 2 (compact dense reference: the tuned two-pair kernels, variant 0; the plain kernel, variant 1) == zmode 0 (the
    reference's materialised dPwn_dzm / dPwn_dTwc) on the same window, and both equal the reference's golden system."""
    import como_amd.odom.backend.photo as photo
    from como_amd.odom.backend.dense_ref import dense_reference_factored
    from tests.test_gpu_hotpath import _ba_call
    G = load_golden(name)
    H0, g0, err0, _ = _ba_call(G)
    B, n, m, Hh, Ww, pixidx, dl_m = _compact_inputs(G)
    dt = G["kf_poses"].dtype
    Kt = dev(G["Knm_Kmminv"].reshape(B, Hh * Ww, m))
    Pw, dl, _, _, _ = dense_reference_factored(dev(G["logzm"]), dev(G["kf_poses"]), Kt, dev(pixidx), dev(G["intrinsics"][0]), dev(dl_m),
                                               Ww, compact=True)
    D = G["H_photo"].shape[0]
    H = torch.zeros((D, D), dtype=dt, device=DEV)
    g = torch.zeros(D, dtype=dt, device=DEV)
    e = torch.zeros((), dtype=torch.float64, device=DEV)
    rid, tid = G["kf_ref_ids"].tolist(), G["kf_target_ids"].tolist()
    table = photo.PairTable(rid, tid, [False] * len(rid), B, dev(G["kf_inds"]), dev(G["recent_inds"]), dev(G["landmark_inds"]),
                            3 * Hh * Ww, 0, DEV)
    monkeypatch.setattr(photo, "BLOCK_VARIANT", variant)
    photo.photo_system_factored(table, poses_all=dev(G["kf_poses"]), aff_all=dev(G["kf_aff_params"].reshape(B, 2)), Pwn=Pw,
                                vals=dev(G["vals_n"].reshape(B, n)), dPwn_dTwc=dl, uvec=None, Kt=Kt, pixidx=dev(pixidx),
                                invz=dev(G["dlogzm_dzm"][:, :, 0, 0]), dzdP=dev(G["dzm_dPwm"][:, 0, 0, :]),
                                img_base=dev(G["kf_img_and_grads"]), K=dev(G["intrinsics"][0]), H_img=Hh, W_img=Ww, H=H, g=g, err_out=e)
    report("ba_compact", case=name, variant=variant, H_rel=rel_err(H, H0), g_rel=rel_err(g, g0), H_rel_ref=rel_err(H, G["H_photo"]),
           H_scaled_ref=scaled_err(H, G["H_photo"]), err=e, err0=err0)
    assert rel_err(H, H0) < tol and rel_err(g, g0) < tol
    assert rel_err(H, G["H_photo"]) < tol * 10 and scaled_err(H, G["H_photo"]) < tol * 100
    assert abs(float(e) - float(err0)) / float(err0) < tol


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-11), (torch.float32, 2e-4)])
def test_small_spd_conditioning(dt, tol):
    """csrc/smallsolve.hip against torch.linalg: factor, inverse, solve (several right-hand-side sweeps), info; forward / backward
    substitution with many columns.  Matrices with the conditioning of a DepthCov K_mm + 1e-6 I would compare two LAPACKs' round-off
    (see test_prep_predictor_vs_golden); here cond ~ 1e3 so that the tolerance measures the kernels."""
    from como_amd.utils.lin_alg import chol_small, cholesky_solve_many, trsm_lower
    g = torch.Generator().manual_seed(3)
    worst = {}
    # (float64, 8 <= n <= 64: the matrix-core tile kernel of csrc/chol.hip, identity-padded to 32 / 64; otherwise the LDS kernel)
    for B, n, k in ((1, 5, 1), (3, 16, 3), (1, 32, 2), (2, 33, 9), (2, 48, 1), (2, 64, 11), (1, 70, 1), (1, 80, 9)):
        M = torch.randn((B, n, n + 8), generator=g, dtype=torch.float64)
        A = (M @ M.mT / (n + 8) + 0.05 * torch.eye(n, dtype=torch.float64)).to(dt)
        rhs = torch.randn((B, n, k), generator=g, dtype=torch.float64).to(dt)
        out = chol_small(dev(A), want_L=True, want_inv=True, rhs=dev(rhs), want_info=True)
        Lr = torch.linalg.cholesky(A.double())
        eye = torch.eye(n, dtype=torch.float64).expand(B, n, n)
        for key, ref in (("L", Lr), ("inv", torch.cholesky_solve(eye, Lr)), ("X", torch.cholesky_solve(rhs.double(), Lr))):
            worst[key] = max(worst.get(key, 0.0), rel_err(out[key], ref))
        assert out["info"].tolist() == [0] * B
        assert torch.equal(out["inv"], out["inv"].mT)                               # exactly symmetric
        assert torch.equal(torch.triu(out["L"], 1), torch.zeros_like(out["L"]))
    # 2-D input, solve only
    A2 = A[0]
    x = chol_small(dev(A2), want_L=False, rhs=dev(rhs[0]))["X"]
    worst["X2"] = rel_err(x, torch.cholesky_solve(rhs[0].double(), torch.linalg.cholesky(A2.double())))
    # info: the order of the first non-positive leading minor (torch.linalg.cholesky_ex)
    Abad = torch.eye(6, dtype=dt)
    Abad[3, 3] = -1.0
    info = chol_small(dev(Abad[None]), want_L=True, want_info=True)["info"]
    assert info.tolist() == torch.linalg.cholesky_ex(Abad[None].double()).info.tolist() == [4]
    for n, where in ((40, 3), (40, 35), (64, 64), (20, 9)):                         # both tiles of the padded 64 x 64 block
        Abad = torch.eye(n, dtype=dt) * 2.0
        Abad[where - 1, where - 1] = -1.0
        info = chol_small(dev(Abad[None]), want_L=True, want_info=True)["info"]
        assert info.tolist() == torch.linalg.cholesky_ex(Abad[None].double()).info.tolist() == [where]
    # many right-hand sides
    for B, n, d in ((1, 64, 3001), (2, 37, 515)):
        M = torch.randn((B, n, n + 8), generator=g, dtype=torch.float64)
        A = M @ M.mT / (n + 8) + 0.05 * torch.eye(n, dtype=torch.float64)
        L = torch.linalg.cholesky(A).to(dt)
        Bm = torch.randn((B, n, d), generator=g, dtype=torch.float64).to(dt)
        worst["trsm"] = max(worst.get("trsm", 0.0), rel_err(trsm_lower(dev(L), dev(Bm)), torch.linalg.solve_triangular(L.double(), Bm.double(), upper=False)))
        worst["trsmT"] = max(worst.get("trsmT", 0.0), rel_err(trsm_lower(dev(L), dev(Bm), trans=True),
                                                               torch.linalg.solve_triangular(L.double().mT, Bm.double(), upper=True)))
        worst["potrs"] = max(worst.get("potrs", 0.0), rel_err(cholesky_solve_many(dev(Bm), dev(L)), torch.cholesky_solve(Bm.double(), L.double())))
    report("small_spd", dtype=str(dt), **worst)
    assert max(worst.values()) < tol
    # the distillation's normal equations (host-side Gram + this solve) against a QR least-squares solution
    if dt == torch.float64:
        from como_amd.depth_cov.core.distill_depth import lstsq_chol
        A = torch.randn((1, 4096, 7), generator=g, dtype=torch.float64)
        b = torch.randn((1, 4096, 1), generator=g, dtype=torch.float64)
        assert rel_err(lstsq_chol(dev(A), dev(b)), torch.linalg.lstsq(A, b).solution) < 1e-9


@pytest.mark.parametrize("ranks", [1, 4, 8])
def test_sharded_exchange_nonfinite_shares_poison_the_system(ranks):
    """Multi-GPU reduce -> all-reduce -> expand (ba_reduce_assemble MODE 1 / MODE 2): a non-finite per-pair sum must reach the
    solver as NaN however many ranks carry it.  `ranks` identical shares are emulated by multiplying the integer exchange buffer
    (an integer all-reduce(sum) of identical buffers).  With the former additive sentinel (2^62 in the summed value) 4 and 8
    ranks wrapped to exactly 0 and the photometric blocks silently vanished."""
    import como_amd.odom.backend.photo as photo
    from como_amd import _lib
    from como_amd.odom.backend.dense_ref import dense_reference_factored
    G = load_golden("ba_window_f64.npz")
    B, n, m, Hh, Ww, pixidx, dl_m = _compact_inputs(G)
    Kt = dev(G["Knm_Kmminv"].reshape(B, Hh * Ww, m))
    Pw, dl, _, _, _ = dense_reference_factored(dev(G["logzm"]), dev(G["kf_poses"]), Kt, dev(pixidx), dev(G["intrinsics"][0]), dev(dl_m),
                                               Ww, compact=True)
    rid, tid = G["kf_ref_ids"].tolist(), G["kf_target_ids"].tolist()
    table = photo.PairTable(rid, tid, [False] * len(rid), B, dev(G["kf_inds"]), dev(G["recent_inds"]), dev(G["landmark_inds"]),
                            3 * Hh * Ww, 0, DEV)
    D = G["H_photo"].shape[0]
    L = _lib.lib()
    plane = int(L.como_sys_fix_plane_elems(D))

    def run(vals):
        sysfix = torch.zeros(2 * plane, dtype=torch.int64, device=DEV)
        seen = {}

        def exchange(t):
            seen["flags"] = t.view(table.b, 3936, 2)[:, 3921:3929].clone()
            t.mul_(ranks)

        photo.photo_system_factored(table, poses_all=dev(G["kf_poses"]), aff_all=dev(G["kf_aff_params"].reshape(B, 2)), Pwn=Pw,
                                    vals=vals, dPwn_dTwc=dl, uvec=None, Kt=Kt, pixidx=dev(pixidx), invz=dev(G["dlogzm_dzm"][:, :, 0, 0]),
                                    dzdP=dev(G["dzm_dPwm"][:, 0, 0, :]), img_base=dev(G["kf_img_and_grads"]), K=dev(G["intrinsics"][0]),
                                    H_img=Hh, W_img=Ww, H=None, g=None, err_out=None, sysfix=sysfix, fix_plane=plane, D=D,
                                    reduce_blocks=exchange, ws={})
        H = torch.zeros((D, D), dtype=torch.float64, device=DEV)
        g = torch.zeros(D, dtype=torch.float64, device=DEV)
        e8 = torch.zeros(8, dtype=torch.float64, device=DEV)
        _lib.check(L.como_sys_finalize(sysfix.data_ptr(), plane, D, H.data_ptr(), g.data_ptr(), e8.data_ptr(), _lib.stream_ptr(torch.device(DEV))),
                   "como_sys_finalize")
        return H, g, seen["flags"]

    vals = dev(G["vals_n"].reshape(B, n)).clone()
    H, g, flags = run(vals)
    assert int(flags.abs().sum()) == 0 and bool(torch.isfinite(H).all())
    assert rel_err(H, ranks * G["H_photo"]) < 1e-9                                  # `ranks` identical finite shares add up
    vals[1, 7] = float("nan")                                                        # one reference intensity: every pair of keyframe 1
    H, g, flags = run(vals)
    assert int(flags.sum()) > 0                                                      # counted, per workgroup of the reduce kernel
    assert bool(torch.isnan(H[0, 0]))                                                # poisoned: the factorisation will report it


def test_tracking_level_record_starts_as_initial_pose():
    """como_track_level_f32 with max_iter = 0 iterations worth of work cannot be requested; instead: a level whose every pixel is
    masked out still leaves a well-defined record -- pose / affine parameters of the caller (never uninitialised memory)."""
    import como_amd.odom.frontend.photo_tracking as pt
    from como_amd import synth
    from como_amd.utils import image_processing as ip
    tp = synth.make_tracking_pair(H=96, W=128, dtype=torch.float32, device=DEV, seed=3, levels=1)
    K = tp["intrinsics"]
    stack = ip.img_and_grads(tp["img_ref"])
    v, u = torch.meshgrid(torch.arange(96., device=DEV), torch.arange(128., device=DEV), indexing="ij")
    ray = torch.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)), -1).reshape(-1, 3)
    P = (tp["depth_ref"][0, 0].reshape(-1, 1) * ray)[None].contiguous()
    vals = tp["img_ref"].reshape(1, -1, 1).contiguous()
    dI = torch.stack((stack[0, 1].reshape(-1), stack[0, 2].reshape(-1)), -1)[None, :, None, :].contiguous()
    J = pt.precalc_jacobians(dI, P, vals, K)
    aff0 = torch.tensor([[[0.01], [-0.02]]], device=DEV)
    term = {"max_iter": 5, "delta_norm": 1e-4, "rel_tol": 1e-4, "grad_norm": 1e-2}
    res = pt.photo_level_tracking_fused(tp["Tji_init"].reshape(1, 4, 4).contiguous(), aff0, vals, P, J, tp["img_cur"], K, term, None)
    assert res is not None
    rec = res[2].cpu()
    assert int(rec[104]) == 0 and int(rec[105]) >= 1                                 # ran, no barrier time-out
    # the same launch again must give the same bits (the record is fully written by the kernel, nothing stale)
    res2 = pt.photo_level_tracking_fused(tp["Tji_init"].reshape(1, 4, 4).contiguous(), aff0, vals, P, J, tp["img_cur"], K, term, None)
    assert torch.equal(res[0], res2[0]) and torch.equal(res[1], res2[1])


@pytest.mark.parametrize("pix", [torch.float64, torch.float32])
def test_band_median_equals_full_depth_pass(pix):
    """Mapping.store_vars' full-image median through `como_depth_band_*` (only the pixels that can still straddle the median are
    re-evaluated) against the depth-only pass over every row of K~, over several GN iterations of the same window (nonmax
    window 4: the reference's operating point): identical medians (float64: the same bits -- both evaluate a candidate with the
    same scalar loop; float32: the full pass forms the dot products on the matrix cores), and most pixels are NOT re-evaluated."""
    import copy
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.backend import dense_ref
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG

    def predictor(cov, cm):
        Kinv, L, Kt = prep_predictor(cov.double(), cm.double(), 1.0)
        return Kinv, L, Kt.to(pix)
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 4
    meds = {}
    for band in (True, False):
        st = synth.make_window(B=4, H=192, W=256, m=64, dtype=torch.float64, device=DEV, seed=5, predictor=predictor)
        old = dense_ref.BAND_MEDIAN
        dense_ref.BAND_MEDIAN = band
        try:
            wb = WindowBA(st, cfg=cfg, pix_dtype=pix, window_full=True)
            seq = []
            for _ in range(6):
                wb.iterate()
                seq.append(wb.median_depths.clone())
            if band:
                key = [k for k in wb.w["dr_ws"] if k[0] == "full"][0]
                bs = wb.w["dr_ws"][key]["band"]
                frac = float(bs["ncand"].sum()) / (bs["calls"] * 4 * 192 * 256)
        finally:
            dense_ref.BAND_MEDIAN = old
        meds[band] = torch.stack(seq).cpu()
    d = (meds[True] - meds[False]).abs().max().item()
    report("band_median", pix=str(pix), max_abs_diff=d, candidate_fraction=frac, medians=meds[True][-1])
    if pix == torch.float64:
        assert torch.equal(meds[True], meds[False])
    else:
        assert d < 2e-6 * float(meds[False].abs().max())
    assert frac < 0.6                                    # (1/6 of the calls build the state: every pixel a candidate there)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_tracker_glue_kernels_vs_torch_formulas(dt):
    """csrc/trackref.hip against the torch formulas they replace (como_amd/odom/Tracking.py mirrors of the reference's
    update_kf_reference / get_reproj_last_kf): reference points, validity masks, inverse-compositional Jacobians of a pyramid
    level; reprojected depth image with the last-point-wins rule, its hit count and exact median."""
    from como_amd import _lib
    from como_amd.geometry.camera import backprojection
    from como_amd.geometry.lie_algebra import se3_exp
    from como_amd.odom.Tracking import _in_image, _project, _rigid
    from como_amd.odom.frontend.photo_tracking import precalc_jacobians
    from como_amd.utils.coords import fill_image, get_test_coords, swap_coords_xy
    from como_amd.utils.select import masked_median
    g = torch.Generator().manual_seed(11)
    b, h, w = 2, 60, 80
    n = h * w
    K = torch.tensor([[70.0, 0, 39.5], [0, 70.0, 29.5], [0, 0, 1]], dtype=dt)
    depth = (1.0 + torch.rand((b, 1, h, w), generator=g, dtype=torch.float64)).to(dt)
    depth[0, 0, 5, 7] = 1e-5                                                      # below the depth threshold
    xi = torch.tensor([[0.02, -0.03, 0.01, 0.4, -0.2, 0.1], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]], dtype=torch.float64)
    rel = se3_exp(xi).to(dt)
    dI = torch.randn((b, n, 1, 2), generator=g, dtype=torch.float64).to(dt)
    vals = torch.rand((b, n, 1), generator=g, dtype=torch.float64).to(dt)
    # ---- torch formulas (the mirror path)
    coords = get_test_coords((h, w), device="cpu", batch_size=b)
    P, _ = backprojection(K, swap_coords_xy(coords), depth[:, 0].reshape(b, n, 1))
    P_all = _rigid(rel, P)
    p_all = _project(K, P_all)
    mask_ref = _in_image(p_all, P_all[:, :, 2:3], (h, w), 50, 1e-4, strict=False)
    J_ref = precalc_jacobians(dI, P_all, vals, K)
    # ---- kernel
    L = _lib.lib()
    Po = torch.empty((b, n, 3), dtype=dt, device=DEV)
    mo = torch.empty((b, n), dtype=torch.uint8, device=DEV)
    Jo = torch.empty((b, n, 1, 8), dtype=dt, device=DEV)
    fn = getattr(L, "como_track_reference_" + _lib.suffix(dt))
    d_depth, d_rel, d_K, d_dI, d_vals = dev(depth), dev(rel).contiguous(), dev(K), dev(dI), dev(vals)     # (held: raw pointers below)
    _lib.check(fn(d_depth.data_ptr(), d_rel.data_ptr(), d_K.data_ptr(), d_dI.data_ptr(), d_vals.data_ptr(), b, h, w, 50.0, 1e-4,
                  Po.data_ptr(), mo.data_ptr(), Jo.data_ptr(), _lib.stream_ptr(torch.device(DEV))), "como_track_reference")
    tol = 1e-12 if dt == torch.float64 else 2e-6
    mm = int((mo.cpu().bool() != mask_ref).sum())
    report("track_reference", dtype=str(dt), P=rel_err(Po, P_all), J=rel_err(Jo, J_ref), mask_mismatch=mm, masked=int((~mask_ref).sum()))
    assert rel_err(Po, P_all) < tol and rel_err(Jo, J_ref) < tol * 20 and mm == 0 and int((~mask_ref).sum()) >= 1
    # ---- reprojection of the newest keyframe's points into a moved frame
    Tck = se3_exp(torch.tensor([[0.01, 0.02, -0.015, 0.08, -0.05, 0.03]], dtype=torch.float64)).to(dt)
    Plast = P_all[-1]
    Pc = _rigid(Tck, Plast[None])
    pc = _project(K, Pc)
    ok = _in_image(pc, Pc[:, :, 2:3], (h, w), 0, 0.0, strict=True)
    img_ref = fill_image(swap_coords_xy(pc)[ok, :], Pc[:, :, 2:3][ok, :], (h, w))
    seen_ref = ~torch.isnan(img_ref)
    order = torch.zeros(h * w, dtype=torch.int64, device=DEV)
    zb = torch.empty(n, dtype=dt, device=DEV)
    img = torch.empty(h * w, dtype=dt, device=DEV)
    seen = torch.empty(h * w, dtype=torch.uint8, device=DEV)
    ns = torch.zeros(1, dtype=torch.int32, device=DEV)
    fr = getattr(L, "como_reproject_depth_" + _lib.suffix(dt))
    d_Tck, d_P = dev(Tck.reshape(4, 4).contiguous()), dev(Plast.contiguous())
    for _ in range(2):                                                           # twice: the claim table must come back clean
        _lib.check(fr(d_Tck.data_ptr(), d_K.data_ptr(), d_P.data_ptr(), n, h, w, order.data_ptr(),
                      zb.data_ptr(), img.data_ptr(), seen.data_ptr(), ns.data_ptr(), _lib.stream_ptr(torch.device(DEV))), "como_reproject_depth")
    sm = int((seen.cpu().bool().view(1, h, w) != seen_ref).sum())
    both = seen.cpu().bool().view(1, h, w) & seen_ref
    dv = (img.cpu().view(1, h, w)[both] - img_ref[both]).abs().max().item()
    med = masked_median(img, seen)
    report("reproject_depth", dtype=str(dt), seen=int(ns[0]), seen_ref=int(seen_ref.sum()), seen_mismatch=sm, value_diff=dv, median=med)
    assert int(order.abs().sum()) == 0 and int(ns[0]) == int(seen.sum())
    assert sm <= (0 if dt == torch.float64 else 2) and dv < tol * 10 and int(seen_ref.sum()) > 1000
    if sm == 0:
        assert abs(float(med) - float(torch.median(img_ref[seen_ref]))) < tol * 10
