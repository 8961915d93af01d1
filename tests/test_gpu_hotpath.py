"""Parity of the HIP hot path (through the C ABI) with the golden vectors and the CPU oracle.

Bars: validity masks and everything that feeds them bit-exact; floating point within the tolerance
written next to each assert.  Run with `-m gpu` on an MI355X.
"""
import numpy as np
import pytest
import torch

from tests.conftest import load_golden, rel_err, report, scaled_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(t):
    return t.to(DEV).contiguous() if torch.is_tensor(t) else t


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 7, 1000, 250_003])
def test_exact_median_select(dtype, n):
    from como_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(n)
    r = torch.randn(n, generator=g, dtype=dtype) * torch.exp(3 * torch.randn(n, generator=g, dtype=dtype))
    r[::5] = 0.25          # many ties
    valid = (torch.rand(n, generator=g) < 0.7)
    valid[0] = True
    rd, vd = dev(r), dev(valid.to(torch.uint8))
    hists = torch.empty(L.como_select_workspace_bytes() // 4, dtype=torch.int32, device=DEV)
    out = torch.empty(3, dtype=dtype, device=DEV)
    sfx = _lib.suffix(dtype)
    s = _lib.stream_ptr()
    _lib.check(L.como_select_begin(hists.data_ptr(), 1, s), "begin")
    for p in range(3 if dtype == torch.float32 else 6):
        _lib.check(getattr(L, "como_select_hist_" + sfx)(rd.data_ptr(), vd.data_ptr(), n, 1, hists.data_ptr(), p, s), "hist")
    _lib.check(getattr(L, "como_select_finish_" + sfx)(hists.data_ptr(), 1, out.data_ptr(), s), "finish")
    ref = torch.median(r[valid].abs())
    o = out.cpu()
    report("select", dtype=str(dtype), n=n, got=o[0], want=ref, nvalid=o[2])
    assert o[0].item() == ref.item()                       # exact
    assert int(o[2].item()) == int(valid.sum())
    assert o[1].item() == (torch.tensor(1.4826, dtype=dtype) * ref).item()


# ------------------------------------------------------------------------------------------------
def _tracking_inputs(T, l):
    return (dev(T["Tji_init"]), dev(T[f"P_l{l}"]), dev(T[f"K_l{l}"]), dev(T[f"cur_l{l}"]),
            torch.zeros((1, 2, 1), device=DEV), dev(T[f"vals_l{l}"]), dev(T[f"J_l{l}"].clone()))


def test_tracking_iter_vs_golden():
    import como_amd.odom.frontend.photo_tracking as pt
    T = load_golden("tracking_f32.npz")
    l = 2
    Tji, P, K, img, aff, vals, J = _tracking_inputs(T, l)
    A_norm = 1.0 / torch.as_tensor((img.shape[-1], img.shape[-2]), device=DEV, dtype=torch.float32)
    Tn, an, delta, mse, gn, pj, valid, depth = pt.tracking_iter(Tji, P, K, img, aff, vals, J, 0.1, A_norm)
    report("tracking_iter", mask_mismatch=(valid.cpu() != T["it_valid"]).sum(), pj_mismatch=(pj.cpu() != T["it_pj"]).sum(),
           depth_mismatch=(depth.cpu() != T["it_depth"]).sum(), delta_rel=rel_err(delta, T["it_delta"]),
           T_err=(Tn.cpu() - T["it_T"]).abs().max(), mse=mse, mse_ref=T["it_mse"], gn=gn, gn_ref=T["it_grad_norm"])
    assert torch.equal(valid.cpu(), T["it_valid"])                    # mask bit-exact
    assert torch.equal(pj.cpu(), T["it_pj"])                          # projected coordinates bit-exact
    assert torch.equal(depth.cpu(), T["it_depth"])
    assert rel_err(delta, T["it_delta"]) < 2e-4                       # fp32 8x8 solve
    assert (Tn.cpu() - T["it_T"]).abs().max() < 1e-5
    assert (an.cpu() - T["it_aff"]).abs().max() < 1e-5
    assert abs(mse.item() - T["it_mse"].item()) < 1e-4 * T["it_mse"].item()
    assert abs(gn.item() - T["it_grad_norm"].item()) < 1e-4 * T["it_grad_norm"].item()
    # the in-place side effect of the reference: dI_dT[..., 6] = -e^{-a} I_j
    assert torch.isfinite(J[..., 6]).all()


def test_tracking_pyramid_vs_golden():
    import como_amd.odom.frontend.photo_tracking as pt
    T = load_golden("tracking_f32.npz")
    term = {"max_iter": 50, "delta_norm": 1e-3, "rel_tol": 1e-3, "grad_norm": 1.0}
    Tf, af = pt.photo_tracking_pyr(dev(T["Tji_init"]), torch.zeros((1, 2, 1), device=DEV),
                                   [dev(T[f"vals_l{i}"]) for i in range(3)], [dev(T[f"P_l{i}"]) for i in range(3)],
                                   [dev(T[f"J_l{i}"].clone()) for i in range(3)], [dev(T[f"mask_l{i}"]) for i in range(3)],
                                   [dev(T[f"K_l{i}"]) for i in range(3)], [dev(T[f"cur_l{i}"]) for i in range(3)], 0.1, term)
    report("tracking_pyr", T_err=(Tf.cpu() - T["pyr_T"]).abs().max(), aff_err=(af.cpu() - T["pyr_aff"]).abs().max(),
           gt_err=(Tf.cpu() - T["Tji_gt"]).abs().max())
    assert (Tf.cpu() - T["pyr_T"]).abs().max() < 1e-4                 # pose within 1e-4 of the reference
    assert (af.cpu() - T["pyr_aff"]).abs().max() < 1e-4


def test_tracking_iter_vs_oracle_fullsize():
    """640x480, N = 307,200: against the oracle and the reference scalars stored in fullsize_scalars.npz."""
    import como_amd.odom.frontend.photo_tracking as pt
    from como_amd import synth
    from oracle import image as oimg, tracking as otrk
    tp = synth.make_tracking_pair(H=480, W=640, dtype=torch.float32, seed=3, levels=1)
    K = tp["intrinsics"]
    gx, gy = oimg.scharr(tp["img_ref"])
    v, u = torch.meshgrid(torch.arange(480.), torch.arange(640.), indexing="ij")
    z = tp["depth_ref"][0, 0]
    P = torch.stack(((u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z), -1).reshape(-1, 3)
    # same construction as the reference's backprojection: z * ray (camera.py:43-54)
    ray = torch.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)), -1).reshape(-1, 3)
    P = z.reshape(-1, 1) * ray
    vals = tp["img_ref"].reshape(-1)
    J = otrk.ic_jacobians(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1), P, vals, K)
    o = otrk.tracking_iter(tp["Tji_init"][0], P, K, tp["img_cur"][0, 0], torch.zeros(2), vals, J)
    out = pt.tracking_iter(dev(tp["Tji_init"]), dev(P[None]), dev(K), dev(tp["img_cur"]), torch.zeros((1, 2, 1), device=DEV),
                           dev(vals[None, :, None]), dev(J[None, :, None, :].clone()), 0.1, None)
    Tn, an, delta, mse, gn, pj, valid, depth = out
    F = load_golden("fullsize_scalars.npz")
    report("tracking_full", mask_mismatch=(valid[0].cpu() != o["valid"]).sum(), nvalid=valid.sum(), nvalid_ref=F["trk_nvalid"],
           delta_rel=rel_err(delta[0, :, 0], o["delta"]), delta_rel_ref=rel_err(delta[0, :, 0], F["trk_delta"][0, :, 0]),
           T_err=(Tn[0].cpu() - o["T"]).abs().max())
    assert torch.equal(valid[0].cpu(), o["valid"])
    assert torch.equal(pj[0, :, 0].cpu(), o["u"]) and torch.equal(pj[0, :, 1].cpu(), o["v"])
    assert rel_err(delta[0, :, 0], o["delta"]) < 5e-4
    assert int(valid.sum()) == int(F["trk_nvalid"])
    assert rel_err(delta[0, :, 0], F["trk_delta"][0, :, 0]) < 5e-4
    assert (Tn[0].cpu() - F["trk_T"][0]).abs().max() < 1e-5


# ------------------------------------------------------------------------------------------------
def _ba_call(G, dt_h=None):
    import como_amd.odom.backend.photo as photo
    dt = G["kf_poses"].dtype
    rid, tid = G["kf_ref_ids"].long(), G["kf_target_ids"].long()
    D = G["H_photo"].shape[0]
    H = torch.zeros((D, D), dtype=dt_h or dt, device=DEV)
    g = torch.zeros(D, dtype=dt_h or dt, device=DEV)
    err = photo.batch_photo_cost(dev(G["vals_n"][rid]), dev(G["kf_aff_params"][rid]), dev(G["Pwn"][rid]), dev(G["kf_poses"][tid]),
                                 dev(G["kf_aff_params"][tid]), dev(G["kf_img_and_grads"][tid]), dev(G["dPwn_dTwc"][rid]),
                                 dev(G["dPwn_dzm"][rid]), dev(G["dzm_dPwm"][rid]), dev(G["kf_inds"][rid]), dev(G["kf_inds"][tid]),
                                 dev(G["landmark_inds"][rid]), dev(G["intrinsics"][0]), H, g)
    return H, g, err, photo.last_aux


@pytest.mark.parametrize("name,tol", [("ba_window_f64.npz", 1e-9), ("ba_window_f32.npz", 3e-4)])
def test_batch_photo_cost_vs_golden(name, tol):
    G = load_golden(name)
    H, g, err, aux = _ba_call(G)
    valid = aux["valid"].cpu()
    sig = aux["sigma"].cpu()
    report("ba_golden", case=name, mask_mismatch=(valid != G["kfpair_valid"]).sum(), pj_mismatch=(aux["pj"].cpu() != G["kfpair_pj"]).sum(),
           r_err=(aux["r"].cpu() - G["kfpair_r"]).abs().max(), sigma=sig[0], sigma_ref=G["sigma_r"], nvalid=sig[1],
           H_rel=rel_err(H, G["H_photo"]), g_rel=rel_err(g, G["g_photo"]), err=err, err_ref=G["photo_err"],
           H_asym=(H - H.T).abs().max())
    assert torch.equal(valid, G["kfpair_valid"])                       # masks bit-exact
    assert torch.equal(aux["pj"].cpu(), G["kfpair_pj"])                # projected coordinates bit-exact
    assert (aux["r"].cpu() - G["kfpair_r"]).abs().max() < (1e-12 if tol < 1e-6 else 1e-5)
    assert abs(sig[0].item() - G["sigma_r"].item()) <= (1e-12 if tol < 1e-6 else 2e-6)
    assert int(sig[1].item()) == int(G["kfpair_valid"].sum())
    assert rel_err(H, G["H_photo"]) < tol and rel_err(g, G["g_photo"]) < tol
    assert scaled_err(H, G["H_photo"]) < tol * 10                      # every block relative to its own scale
    assert abs(err.item() - G["photo_err"].item()) / G["photo_err"].item() < tol


def test_gram_layout_vs_oracle():
    """Raw per-pair 80x80 Gram blocks against the oracle's (asymmetric -> catches any MFMA row/col mix-up)."""
    from oracle import photo_ba
    import como_amd.odom.backend.photo as photo
    G = load_golden("ba_window_f64.npz")
    rid, tid = G["kf_ref_ids"].long(), G["kf_target_ids"].long()
    r, valid, J = photo_ba.pair_rows(G["vals_n"][rid], G["kf_aff_params"][rid], G["Pwn"][rid], G["kf_poses"][tid], G["kf_aff_params"][tid],
                                     G["kf_img_and_grads"][tid], G["dPwn_dTwc"][rid], G["dPwn_dzm"][rid], G["intrinsics"][0])
    sigma = photo_ba.robust_scale(r, valid)
    Gm, gv, err = photo_ba.pair_blocks(r, valid, J, sigma)
    b, n, _, m, _ = G["dPwn_dzm"][rid].shape
    D = G["H_photo"].shape[0]
    H = torch.zeros((D, D), dtype=torch.float64, device=DEV)
    g = torch.zeros(D, dtype=torch.float64, device=DEV)
    e = torch.zeros((), dtype=torch.float64, device=DEV)
    ar = torch.arange(b, dtype=torch.int32, device=DEV)
    aff_all = torch.cat((G["kf_aff_params"][rid].reshape(b, 2), G["kf_aff_params"][tid].reshape(b, 2))).to(DEV)
    Hh, Ww = G["kf_img_and_grads"].shape[-2:]
    photo.linearize(dtype=torch.float64, b=b, n=n, m=m, H_img=Hh, W_img=Ww, zmode=0, Pwn=dev(G["Pwn"][rid]),
                    vals=dev(G["vals_n"][rid].reshape(b, n)), dPwn_dTwc=dev(G["dPwn_dTwc"][rid]), zjac=dev(G["dPwn_dzm"][rid]),
                    poses_all=dev(G["kf_poses"][tid]), aff_all=aff_all, img_base=dev(G["kf_img_and_grads"][tid]),
                    K=dev(G["intrinsics"][0]), ref_slot=ar, ref_aff=ar, tgt_aff=(ar + b).contiguous(), tgt_pose=ar,
                    tgt_img=torch.arange(b, dtype=torch.int64, device=DEV) * (3 * Hh * Ww),
                    pose_ref_inds=dev(G["kf_inds"][rid]), pose_tgt_inds=dev(G["kf_inds"][tid]),
                    landmark_inds=dev(G["landmark_inds"][rid]), dzdP=dev(G["dzm_dPwm"][rid][:, 0, 0, :]), H=H, g=g, err_out=e,
                    want_blocks=True, chunks=3)
    raw = photo.last_aux["blocks"].cpu()                        # (b, 3936)
    got = torch.zeros((b, 80, 80), dtype=torch.float64)
    gg = torch.zeros((b, 80), dtype=torch.float64)

    def col(t, ci):
        return ci if t == 0 else 16 + 4 * ci + (t - 1)
    tt = 0
    for ti in range(5):
        for tj in range(ti, 5):
            for rg in range(4):
                for lane in range(64):
                    ri, ci = (lane >> 4) + 4 * rg, lane & 15     # f64 MFMA layout
                    got[:, col(ti, ri), col(tj, ci)] = raw[:, tt * 256 + rg * 64 + lane]
            tt += 1
    for t in range(5):
        for ci in range(16):
            gg[:, col(t, ci)] = -raw[:, 15 * 256 + t * 16 + ci]
    want = torch.zeros((b, 80, 80), dtype=torch.float64)
    want[:, :16 + m, :16 + m] = Gm
    iu = torch.triu(torch.ones(80, 80, dtype=torch.bool))
    # only the upper block triangle is produced (diagonal blocks are full)
    blk = torch.zeros(80, 80, dtype=torch.bool)
    for ti in range(5):
        for tj in range(ti, 5):
            rows = [col(ti, i) for i in range(16)]
            cols = [col(tj, j) for j in range(16)]
            blk[np.ix_(rows, cols)] = True
    d = ((got - want) * blk).abs().max() / want.abs().max()
    dg = (gg[:, :16 + m] - gv).abs().max() / gv.abs().max()
    report("gram_layout", gram_rel=d, g_rel=dg, err=photo.last_aux["blocks"][:, 15 * 256 + 80].sum(), err_ref=err)
    assert d < 1e-10 and dg < 1e-10


def test_create_photo_system_recent_vs_golden():
    """Keyframe + one-way (recent) pairs in one batch: the whole create_photo_system contract."""
    import como_amd.odom.backend.photo as photo
    from oracle import dense_ref, depthcov
    G = load_golden("ba_window_recent_f64.npz")
    K = G["intrinsics"][0]
    Kt = depthcov.prep_predictor(G["cov_params_img"], G["coords_m"], 1.0)[2]     # inputs only (oracle == golden-checked)
    cn = G["coords_n"]
    bi = torch.arange(cn.shape[0])[:, None].expand(-1, cn.shape[1])
    Pw, dT, dz, med, _ = dense_ref.dense_reference(G["logzm"], G["kf_poses"], Kt[bi, cn[..., 0], cn[..., 1], :], cn, K,
                                                   G["dlogzm_dzm"] @ G["dzm_dTwc"], G["dlogzm_dzm"])
    D = G["H_photo"].shape[0]
    H = torch.zeros((D, D), dtype=torch.float64, device=DEV)
    g = torch.zeros(D, dtype=torch.float64, device=DEV)
    cfg = {"pairwise_batch_size": 128, "radius_thresh": 0.0, "degrees_thresh": 0.0}
    err, kfp, owp = photo.create_photo_system(dev(G["kf_poses"]), dev(G["kf_aff_params"]), dev(G["recent_poses"]), dev(G["recent_aff_params"]),
                                              dev(Pw), dev(dT), dev(dz), dev(G["dzm_dPwm"]), dev(med), dev(G["vals_n"]),
                                              dev(G["kf_img_and_grads"]), dev(G["recent_img_and_grads"]), dev(G["kf_timestamps"]),
                                              dev(G["recent_timestamps"]), dev(G["intrinsics"]), H, g, cfg, dev(G["kf_inds"]),
                                              dev(G["recent_inds"]), dev(G["landmark_inds"]))
    report("ba_recent", H_rel=rel_err(H, G["H_photo"]), g_rel=rel_err(g, G["g_photo"]), err=err, err_ref=G["photo_err"])
    assert kfp[0] == G["kf_ref_ids"].tolist() and owp[1] == G["ow_target_ids"].tolist()
    assert rel_err(H, G["H_photo"]) < 1e-9 and rel_err(g, G["g_photo"]) < 1e-9
    assert abs(float(err) - G["photo_err"].item()) / G["photo_err"].item() < 1e-9


def test_factored_path_matches_reference_signature_path():
    """zmode 1 (rank-1 factors, K~ read in place) == zmode 0 (materialised dPwn_dzm) on the same window."""
    import como_amd.odom.backend.photo as photo
    G = load_golden("ba_window_f64.npz")
    H0, g0, err0, _ = _ba_call(G)
    B, n = G["coords_n"].shape[:2]
    m = G["coords_m"].shape[1]
    K = G["intrinsics"][0]
    cn = G["coords_n"]
    Hh, Ww = G["kf_img_and_grads"].shape[-2:]
    ray = torch.stack(((cn[..., 1].double() - K[0, 2]) / K[0, 0], (cn[..., 0].double() - K[1, 2]) / K[1, 1],
                       torch.ones(B, n, dtype=torch.float64)), -1)
    zn = torch.exp(G["logzn"])
    uvec = torch.einsum("bij,bnj->bni", G["kf_poses"][:, :3, :3], ray * zn)
    invz = G["dlogzm_dzm"][:, :, 0, 0]
    pixidx = (cn[..., 0] * Ww + cn[..., 1]).to(torch.int32)
    D = G["H_photo"].shape[0]
    H = torch.zeros((D, D), dtype=torch.float64, device=DEV)
    g = torch.zeros(D, dtype=torch.float64, device=DEV)
    e = torch.zeros((), dtype=torch.float64, device=DEV)
    rid, tid = G["kf_ref_ids"].tolist(), G["kf_target_ids"].tolist()
    table = photo.PairTable(rid, tid, [False] * len(rid), B, dev(G["kf_inds"]), dev(G["recent_inds"]), dev(G["landmark_inds"]),
                            3 * Hh * Ww, 0, DEV)
    soa = lambda t, k: dev(t.reshape(B, n, k).permute(0, 2, 1))          # fast path takes (B, k, n) planes
    photo.photo_system_factored(table, poses_all=dev(G["kf_poses"]), aff_all=dev(G["kf_aff_params"].reshape(B, 2)), Pwn=soa(G["Pwn"], 3),
                                vals=dev(G["vals_n"].reshape(B, n)), dPwn_dTwc=soa(G["dPwn_dTwc"], 18), uvec=soa(uvec, 3),
                                Kt=dev(G["Knm_Kmminv"].reshape(B, Hh * Ww, m)), pixidx=dev(pixidx), invz=dev(invz),
                                dzdP=dev(G["dzm_dPwm"][:, 0, 0, :]), img_base=dev(G["kf_img_and_grads"]), K=dev(K), H_img=Hh, W_img=Ww,
                                H=H, g=g, err_out=e)
    report("ba_factored", H_rel=rel_err(H, H0), g_rel=rel_err(g, g0), H_rel_ref=rel_err(H, G["H_photo"]), err=e, err0=err0)
    assert rel_err(H, H0) < 1e-10 and rel_err(g, g0) < 1e-10
    assert rel_err(H, G["H_photo"]) < 1e-9


def test_mixed_precision_f32_pixels_f64_system():
    """f32 per-pixel path accumulating into an f64 system (the bench configuration): within fp32 tolerance of the f64 golden."""
    G32 = load_golden("ba_window_f32.npz")
    G64 = load_golden("ba_window_f64.npz")
    H, g, err, _ = _ba_call(G32, dt_h=torch.float64)
    report("ba_mixed", H_rel=rel_err(H, G64["H_photo"]), g_rel=rel_err(g, G64["g_photo"]))
    assert scaled_err(H, G64["H_photo"]) < 3e-4 and rel_err(g, G64["g_photo"]) < 3e-4


# ------------------------------------------------------------------------------------------------
def test_como_backends_vs_golden():
    import como_amd.como_backends as cb
    C = load_golden("cov_ops_f32.npz")
    K12 = cb.cross_covariance(dev(C["x1"]), dev(C["E1"]), dev(C["x2"]), dev(C["E2"]), float(C["scale"]))
    xs, Es = dev(C["xs"]), dev(C["Es"])
    Ks = cb.cross_covariance(xs[:, :5], Es[:, :5], xs[:, 3:], Es[:, 3:], 1.0)          # strided views
    K64 = cb.cross_covariance(dev(C["x1"].double()), dev(C["E1"].double()), dev(C["x2"].double()), dev(C["E2"].double()), float(C["scale"]))
    L, obs, var = dev(C["app_L0"].clone()), dev(C["app_obs0"].clone()), dev(C["app_var0"].clone())
    ax, aE, dn, Ed = dev(C["app_x"]), dev(C["app_E"]), dev(C["app_dn"]), dev(C["app_Ed"])
    for i in range(1, 6):
        k_ni = cb.cross_covariance(ax[:, :i], aE[:, :i], ax[:, i:i + 1], aE[:, i:i + 1], 1.0)
        k_id = cb.cross_covariance(ax[:, i:i + 1], aE[:, i:i + 1], dn, Ed, 1.0)
        assert cb.get_new_chol_obs_info(L, obs, var, k_ni, k_id, 1.0, i) is None
    report("como_backends", K12=rel_err(K12, C["K12"]), Ks=rel_err(Ks, C["K_slice"]), K64=rel_err(K64, C["K12"].double()),
           L=rel_err(L, C["app_L"]), obs=rel_err(obs, C["app_obs"]), var=(var.cpu() - C["app_var"]).abs().max())
    assert rel_err(K12, C["K12"]) < 2e-6 and rel_err(Ks, C["K_slice"]) < 2e-6 and rel_err(K64, C["K12"].double()) < 2e-6
    assert rel_err(L, C["app_L"]) < 1e-5 and rel_err(obs, C["app_obs"]) < 1e-4 and (var.cpu() - C["app_var"]).abs().max() < 1e-5
    with pytest.raises(RuntimeError):
        cb.get_new_chol_obs_info(L.transpose(1, 2), obs, var, k_ni, k_id, 1.0, 1)      # "must be contiguous"
    with pytest.raises(RuntimeError):
        cb.cross_covariance(C["x1"], dev(C["E1"]), dev(C["x2"]), dev(C["E2"]), 1.0)    # mixed devices


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,tol", [("ba_window_f64.npz", 1e-12), ("ba_window_f32.npz", 2e-5)])
def test_dense_reference_factored_vs_golden(name, tol):
    from como_amd.odom.backend.dense_ref import dense_reference_factored
    G = load_golden(name)
    B, n = G["coords_n"].shape[:2]
    m = G["coords_m"].shape[1]
    Hh, Ww = G["kf_img_and_grads"].shape[-2:]
    cn = G["coords_n"]
    pixidx = (cn[..., 0] * Ww + cn[..., 1]).to(torch.int32)
    dl = G["dlogzm_dzm"] @ G["dzm_dTwc"]
    Pwn, dT, uvec, med, logzn = dense_reference_factored(dev(G["logzm"]), dev(G["kf_poses"]), dev(G["Knm_Kmminv"].reshape(B, Hh * Ww, m)),
                                                         dev(pixidx), dev(G["intrinsics"][0]), dev(dl), Ww)
    Pw = Pwn.permute(0, 2, 1).cpu()
    dTa = dT.permute(0, 2, 1).reshape(B, n, 3, 6).cpu()
    # rank-1 reconstruction of the tensor the reference materialises
    dz = torch.einsum("bni,bnk,bk->bnik", uvec.permute(0, 2, 1).cpu(), G["Knm_Kmminv"].reshape(B, -1, m)[torch.arange(B)[:, None], pixidx.long()],
                      G["dlogzm_dzm"][:, :, 0, 0])
    report("dense_ref", case=name, Pwn=rel_err(Pw, G["Pwn"]), Pwn_exact=bool(torch.equal(Pw, G["Pwn"])), dT=rel_err(dTa, G["dPwn_dTwc"]),
           dz=rel_err(dz, G["dPwn_dzm"][..., 0]), med=(med.cpu() - G["median_depths"]).abs().max(), logz=rel_err(logzn.cpu(), G["logzn"][..., 0]))
    assert rel_err(Pw, G["Pwn"]) < tol and rel_err(dTa, G["dPwn_dTwc"]) < tol * 10
    assert rel_err(dz, G["dPwn_dzm"][..., 0]) < tol * 10
    assert (med.cpu() - G["median_depths"]).abs().max() < tol * 10


@pytest.mark.parametrize("name,tol", [("ba_window_f64.npz", 1e-7), ("ba_window_f32.npz", 2e-4)])
def test_prep_predictor_vs_golden(name, tol):
    from como_amd.depth_cov.core.covariance import prep_predictor
    G = load_golden(name)
    Kinv, L, Kt = prep_predictor(dev(G["cov_params_img"]), dev(G["coords_m"]), 1.0)
    report("prep_predictor", case=name, Kinv=rel_err(Kinv, G["K_mm_inv"]), L=rel_err(L, G["L_mm"]), Kt=rel_err(Kt, G["Knm_Kmminv"]))
    # f64: the reference's Python twin casts coordinate differences to float32 (kernels.py:25), so its own K_mm carries
    # ~1e-8 relative noise; the tolerance reflects that, not the kernel.
    assert rel_err(L, G["L_mm"]) < tol and rel_err(Kinv, G["K_mm_inv"]) < tol * 50 and rel_err(Kt, G["Knm_Kmminv"]) < tol * 50


def _window_state(G):
    st = {k: dev(G[k]) for k in ("intrinsics", "kf_poses", "kf_aff_params", "kf_img_and_grads", "coords_m", "correspondence_mask",
                                 "P_m", "kf_timestamps", "obs_ref_mask", "pm_first_obs", "L_mm", "K_mm_inv", "Knm_Kmminv", "pose_anchor", "P_anchor")}
    return st


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("pix", [torch.float64, torch.float32])
def test_window_iterate_vs_golden(pix, fused):
    """One full Mapping.iterate()-equivalent (scaffold -> dense ref -> photo system -> priors -> solve -> update)."""
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    import copy
    G = load_golden("ba_window_f64.npz")
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 2
    wb = WindowBA(_window_state(G), cfg=cfg, pix_dtype=pix, window_full=True, fused=fused)
    wb.median_depths.copy_(dev(G["median_depths_in"]))
    H, g = wb.linearize()
    Hrel, grel = scaled_err(H, G["H_full"]), rel_err(g, G["g_full"])    # Jacobi-scaled: every block relative to its own magnitude
    wb2 = WindowBA(_window_state(G), cfg=cfg, pix_dtype=pix, window_full=True, fused=fused)
    wb2.median_depths.copy_(dev(G["median_depths_in"]))
    delta = wb2.iterate()
    perr = (wb2.kf_poses.cpu() - G["kf_poses_new"]).abs().max().item()
    report("window_iterate", pix=str(pix), fused=fused, H_rel=Hrel, g_rel=grel, delta_rel=rel_err(delta, G["delta"]), pose_err=perr,
           P_err=(wb2.P_m.cpu() - G["P_new"]).abs().max(), info=int(lin_info()))
    tol = 1e-8 if pix == torch.float64 else 3e-4
    assert Hrel < tol and grel < tol
    assert perr < (1e-9 if pix == torch.float64 else 1e-4)           # poses within 1e-4 of the reference
    assert (wb2.P_m.cpu() - G["P_new"]).abs().max() < (1e-8 if pix == torch.float64 else 5e-3)


def lin_info():
    import como_amd.odom.backend.linear_system as ls
    return ls.solve_system.last_info


@pytest.mark.parametrize("D", [5, 63, 64, 65, 200, 760, 1500])
def test_cholesky_solve(D):
    import como_amd.odom.backend.linear_system as ls
    g0 = torch.Generator().manual_seed(D)
    A = torch.randn((D, D + 8), generator=g0, dtype=torch.float64)
    H = A @ A.T + 1e-3 * torch.eye(D, dtype=torch.float64)
    H[0, 0] += 1e12                                      # the pose-anchor scale of the real system
    g = torch.randn(D, generator=g0, dtype=torch.float64)
    ref = torch.cholesky_solve(g[:, None], torch.linalg.cholesky(H))
    d = ls.solve_system(dev(H), dev(g))
    resid = (H @ d.cpu() - g[:, None]).abs().max() / g.abs().max()
    report("chol", D=D, rel=rel_err(d, ref), resid=resid, info=int(ls.solve_system.last_info))
    assert int(ls.solve_system.last_info) == 0
    assert rel_err(d, ref) < 1e-8
    Hbad = H.clone()
    k = D // 2
    Hbad[k, k] = -1.0
    ls.solve_system(dev(Hbad), dev(g))
    assert int(ls.solve_system.last_info) == k + 1        # first non-positive pivot, 1-based (cholesky_ex convention)


# ------------------------------------------------------------------------------------------------
# DepthCov covariance network (a36): HIP layers vs torch CPU ops, then the whole network vs the reference's output.
def test_nn_layers_vs_torch():
    import torch.nn.functional as F
    from como_amd.depth_cov.nn import UNet as U
    g = torch.Generator().manual_seed(3)
    for (cin, cout, k, H, W) in [(3, 16, 3, 20, 28), (16, 32, 3, 6, 8), (48, 16, 1, 9, 7), (32, 3, 1, 12, 16), (64, 64, 3, 3, 4)]:
        x = torch.randn(2, cin, H, W, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g)
        ref = F.conv2d(x, w, b, padding=k // 2)
        got = U._Conv(dev(w), dev(b))(dev(x))
        e = rel_err(got, ref)
        report(f"nn_conv_{cin}_{cout}_{k}", rel_err=e)
        assert e < 2e-6, (cin, cout, k, e)                 # fp32 accumulation order only
    x = torch.randn(2, 32, 10, 12, generator=g) * 3 + 1
    ga, be, res = torch.randn(32, generator=g), torch.randn(32, generator=g), torch.randn(2, 32, 10, 12, generator=g)
    ref = F.leaky_relu(res + F.group_norm(x, 16, ga, be, 1e-5), 0.01)
    got = U._groupnorm(dev(x), dev(ga), dev(be), 2, residual=dev(res))
    assert rel_err(got, ref) < 2e-6
    ref = F.leaky_relu(F.group_norm(x, 16, ga, be, 1e-5), 0.01)
    assert rel_err(U._groupnorm(dev(x), dev(ga), dev(be), 1), ref) < 2e-6
    assert torch.equal(U.maxpool2(dev(x)).cpu(), F.max_pool2d(x, 2))
    up = F.interpolate(x, scale_factor=(2, 2), mode="bilinear", align_corners=False)
    assert rel_err(U.upsample2x(dev(x)), up) < 1e-6
    for dt, tol in ((torch.float32, 2e-6), (torch.float64, 1e-13)):
        img = torch.rand(1, 3, 72, 100, generator=g).to(dt)
        for size in ((32, 64), (72, 100), (150, 211), (19, 23)):
            ref = F.interpolate(img, size=list(size), mode="bilinear", antialias=True, align_corners=False)
            e = rel_err(U.resize_aa(dev(img), size), ref)
            assert e < tol, (dt, size, e)


def test_depthcov_network_vs_reference():
    """DepthCovModule.forward and Mapping.run_model on the seeded weights: HIP network vs the reference's outputs
    (golden) and vs the CPU oracle.  fp32 tolerance 2e-4 relative to the largest covariance entry (30 conv / GroupNorm
    layers deep, MFMA accumulation order)."""
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule, run_model
    from como_amd.synth import depthcov_state_dict
    from oracle import unet as ounet
    g = load_golden("depthcov_net.npz")
    sd_cpu = depthcov_state_dict(int(g["seed"]))
    model = DepthCovModule({k: dev(v) for k, v in sd_cpu.items()})
    covs = model(dev(g["rgb"]))
    assert len(covs) == 4
    for i, c in enumerate(covs):
        e = rel_err(c, g[f"cov_level{i}"])
        report(f"depthcov_level{i}", rel_err=e)
        assert e < 2e-4, (i, e)
    cov = run_model(model, dev(g["rgb_big"]), network_size=g["net_size"].tolist())
    assert cov.dtype == torch.float64 and tuple(cov.shape) == tuple(g["run_model_cov"].shape)
    e = rel_err(cov, g["run_model_cov"])
    report("depthcov_run_model", rel_err=e)
    assert e < 2e-4
    # full operating size of the reference (192x256 network, 480x640 image) against the oracle
    gen = torch.Generator().manual_seed(11)
    rgb = torch.rand(1, 3, 480, 640, generator=gen)
    rgb = torch.nn.functional.avg_pool2d(rgb, 9, stride=1, padding=4)
    with torch.no_grad():
        ref = ounet.run_model(sd_cpu, rgb)
    got = run_model(model, dev(rgb))
    e = rel_err(got, ref)
    report("depthcov_run_model_480x640", rel_err=e)
    assert e < 2e-4


# ------------------------------------------------------------------------------------------------
# Image operators (a37, a20, a22) vs the reference's outputs (golden) and the oracle
def test_image_ops_vs_golden():
    from como_amd.utils import image_processing as ip
    from como_amd.odom.backend import sparse_map as smap
    from oracle import image as oimg
    I = load_golden("image_ops.npz")
    img = I["img"]                                                    # float64 fixture
    for dt, tol in ((torch.float64, 1e-14), (torch.float32, 2e-6)):
        s = ip.img_and_grads(dev(img.to(dt)))
        assert torch.equal(s[:, :1].cpu(), img.to(dt))
        assert rel_err(s[:, 1:2], I["gx"]) < tol and rel_err(s[:, 2:3], I["gy"]) < tol
        gx, gy = ip.ImageGradientModule(1, DEV, dt)(dev(img.to(dt)))
        assert rel_err(gx, I["gx"]) < tol and rel_err(gy, I["gy"]) < tol
        pyr = ip.ImagePyramidModule(1, 0, 3, DEV, dt)(dev(img.to(dt)))
        for i in range(3):
            assert tuple(pyr[i].shape) == tuple(I[f"pyr{i}"].shape)
            assert rel_err(pyr[i], I[f"pyr{i}"]) < tol, (dt, i)
    Kp = ip.IntrinsicsPyramidModule(0, 3, DEV)(dev(I["K"]), [1.0, 1.0])
    for i in range(3):
        assert torch.equal(Kp[i].cpu(), I[f"K{i}"])
    # odd sizes / reflect borders against the oracle
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 37, 53, generator=g, dtype=torch.float64)
    gx, gy = oimg.scharr(x)
    s = ip.img_and_grads(dev(x))
    assert rel_err(s[:, 1:2], gx) < 1e-14 and rel_err(s[:, 2:3], gy) < 1e-14
    assert rel_err(ip.blur_down(dev(x)), oimg.blur_down(x)) < 1e-14
    # sub-selection: bit-exact indices vs the reference's coords_n of the window fixtures, and vs max_pool2d at 480x640
    for name in ("ba_window_f64.npz", "ba_window_f32.npz"):
        G = load_golden(name)
        cn, bi, pix = smap.subselect_pixels(dev(G["kf_img_and_grads"]), 2, want_pixidx=True)
        assert torch.equal(cn.cpu(), G["coords_n"])
        W = G["kf_img_and_grads"].shape[-1]
        assert torch.equal(pix.cpu().long(), G["coords_n"][..., 0] * W + G["coords_n"][..., 1])
    big = torch.rand(2, 3, 480, 640, generator=g, dtype=torch.float32)
    big[:, 1:, 100:140, 200:260] = 0.25                               # flat patch: ties resolve to the first pixel
    for w in (1, 2, 4, 7):
        gn = torch.sqrt(big[:, 1] ** 2 + big[:, 2] ** 2)
        _, idx = torch.nn.functional.max_pool2d(gn[:, None], kernel_size=w, return_indices=True)
        idx = idx.reshape(2, -1)
        cn, _ = smap.subselect_pixels(dev(big), w)
        assert torch.equal(cn.cpu(), torch.stack((idx // 640, idx % 640), dim=-1)), w


def test_precalc_jacobians_vs_golden():
    from como_amd.odom.frontend import photo_tracking as pt
    from oracle import tracking as otrk
    T = load_golden("tracking_f32.npz")
    g = torch.Generator().manual_seed(9)
    N = 5000
    P = torch.rand(1, N, 3, generator=g) * torch.tensor([2.0, 1.5, 3.0]) + torch.tensor([-1.0, -0.75, 0.5])
    dI = torch.randn(1, N, 1, 2, generator=g)
    vals = torch.rand(1, N, 1, generator=g)
    K = T["K"] if "K" in T else torch.tensor([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])
    for dt, tol in ((torch.float32, 2e-6), (torch.float64, 1e-14)):
        J = pt.precalc_jacobians(dev(dI.to(dt)), dev(P.to(dt)), dev(vals.to(dt)), dev(K.to(dt)))
        ref = otrk.ic_jacobians(dI[0, :, 0].to(dt), P[0].to(dt), vals[0, :, 0].to(dt), K.to(dt))
        assert tuple(J.shape) == (1, N, 1, 8)
        assert rel_err(J[0, :, 0], ref) < tol


# ------------------------------------------------------------------------------------------------
# Two-frame SfM system (a28/a29) through the BA kernels vs the reference's construct_photo_system
def test_two_frame_sfm_system_vs_golden():
    from como_amd.odom.frontend import two_frame_sfm as sfm
    S = load_golden("sfm_f64.npz")
    m = S["logz_m"].shape[1]
    D = 6 + m
    H = torch.zeros((D, D), dtype=torch.float64, device=DEV)
    g = torch.zeros((D,), dtype=torch.float64, device=DEV)
    aff = torch.zeros((1, 2, 1), dtype=torch.float64, device=DEV)
    err, log_depth, coords_j, depths_j, valid, Pi = sfm.construct_photo_system(
        dev(S["Tji"]), dev(S["logz_m"]), aff, dev(S["coords_i"]), dev(S["vals_i"]), dev(S["Kt"]), dev(S["img_and_grads_j"]),
        dev(S["K"]), 0.1, H, g)
    mism = (valid.cpu() != S["valid"]).sum().item()
    eH, eg = rel_err(H, S["H"]), rel_err(g, S["g"])
    ee = abs(err.item() - S["err"].item()) / S["err"].item()
    report("sfm_system", mask_mismatch=mism, H_rel=eH, g_rel=eg, err_rel=ee, logz_rel=rel_err(log_depth, S["log_depth"]))
    assert mism == 0                                            # validity mask bit-exact
    assert rel_err(log_depth, S["log_depth"]) < 1e-13 and rel_err(Pi, S["Pi"]) < 1e-13
    assert eH < 1e-10 and eg < 1e-10 and ee < 1e-10            # float64; same float32 A_norm as the reference
    assert coords_j.shape[1] == int(S["valid"].sum()) and depths_j.shape[1] == coords_j.shape[1]
    # a GN step reduces the cost and the loop terminates
    dr, Hd = sfm.linearize_sparse_depth_prior(torch.eye(m, dtype=torch.float64, device=DEV)[None])
    T, d, _, _, _, mld = sfm.two_frame_sfm(dev(S["Tji"]), dev(S["logz_m"]), aff, dev(S["coords_i"]), dev(S["vals_i"]), dev(S["Kt"]),
                                          dev(S["img_and_grads_j"]), dr, Hd, dev(S["K"]), {"photo": 0.1}, {},
                                          {"max_iter": 6, "delta_norm": 1e-9, "rel_tol": 1e-9})
    assert torch.isfinite(T).all() and torch.isfinite(d).all() and sfm.two_frame_sfm.last_iters <= 6


# ------------------------------------------------------------------------------------------------
# Greedy conditional-entropy sampler (a33-a35): chosen pixels identical to the reference's sample_sparse_coords
def test_greedy_sampler_vs_golden():
    from como_amd.depth_cov.core import samplers
    from oracle import depthcov as odc
    C = load_golden("cov_ops_f32.npz")
    num, border, dth = int(C["samp_num"]), int(C["samp_border"]), float(C["samp_dist_thresh"])
    coords, inds = samplers.sample_sparse_coords(dev(C["cov_params_img"]), num, "greedy_conditional_entropy", border=border,
                                                 dist_thresh=dth, signal_var=torch.tensor(1.0))
    assert torch.equal(inds.cpu(), C["samp_domain_inds"]) and torch.equal(coords.cpu(), C["samp_coords"])
    # the step-by-step host loop (taken when early termination is requested; never triggered with this threshold)
    coords2, inds2 = samplers.sample_sparse_coords(dev(C["cov_params_img"]), num, "greedy_conditional_entropy", border=border,
                                                   dist_thresh=dth, signal_var=torch.tensor(1.0), terminate_early=True,
                                                   max_stdev_thresh=-1.0)
    assert torch.equal(inds2.cpu(), C["samp_domain_inds"])
    # full operating point of the reference: 64 points on a 192x256 covariance image, vs the oracle
    from como_amd.synth import synthetic_cov_params
    cov = synthetic_cov_params(1, 96, 128, seed=3, dtype=torch.float64).float()
    idx, pix, *_ = odc.greedy_sampler(cov, 32, 1.0, 2, 0.05)
    coords, inds = samplers.sample_sparse_coords(dev(cov), 32, "greedy_conditional_entropy", border=2, dist_thresh=0.05,
                                                 signal_var=torch.tensor(1.0))
    same = (inds.cpu()[0] == idx).sum().item()
    report("greedy_sampler", same_of_32=same)
    assert same == 32 and torch.equal(coords.cpu()[0], pix)


# ------------------------------------------------------------------------------------------------
# Full-size window (the bench workload: 8 keyframes, 640x480, m = 64): size-independent properties
def _full_window(pix_dtype, window, seed=0):
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    import copy

    def predictor(cov, cm):
        Kinv, L, Kt = prep_predictor(cov.double(), cm.double(), 1.0)
        return Kinv, L, Kt.to(pix_dtype)
    st = synth.make_window(B=8, H=480, W=640, m=64, dtype=torch.float64, device=DEV, seed=seed, predictor=predictor)
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = window
    return WindowBA(st, cfg=cfg, pix_dtype=pix_dtype, window_full=True), st


@pytest.mark.parametrize("window", [4, 1])
def test_fullsize_window_properties(window):
    """(1) the float32 pixel path agrees with the float64 (reference mapping dtype) pixel path: first-iteration system and
    pose update within the stated 1e-4; (2) H symmetric; (3) Gauss-Newton converges to the ground-truth poses and the
    cost settles below its starting value; (4) the captured-graph replay equals the eager iteration."""
    w32, st = _full_window(torch.float32, window)
    w64, _ = _full_window(torch.float64, window)
    H32, g32 = w32.linearize()
    H64, g64 = w64.linearize()
    eH, eg = rel_err(H32, H64), rel_err(g32, g64)
    sym = ((H64 - H64.T).abs().max() / H64.abs().max()).item()
    d32, d64 = w32.iterate(), w64.iterate()                      # (re-linearises; same state)
    upd = (w32.kf_poses - w64.kf_poses).abs().max().item()
    report("fullsize_window", window=window, H_rel_f32_vs_f64=eH, g_rel=eg, sym=sym, pose_update_diff=upd,
           n=w32.n, D=w32.dim)
    assert eH < 2e-5 and eg < 2e-4 and sym < 1e-12
    assert upd < 1e-4
    gt = st["poses_gt"]
    e0 = (w32.kf_poses - gt).abs().max().item()
    errs = []
    for _ in range(8):
        w32.iterate()
        errs.append(float(w32.err))
    e1 = (w32.kf_poses - gt).abs().max().item()
    report("fullsize_window_convergence", window=window, pose_err_after_1=e0, pose_err_after_9=e1, cost=errs)
    assert e1 < 5e-4 and e1 < e0
    # the robust scale is re-estimated every iteration and the cost is a float32 running sum: settled = flat to 1e-3
    assert errs[-1] < errs[0] and all(abs(e - errs[-1]) <= 1e-3 * errs[-1] for e in errs[2:])
    # graph replay == eager: two identical copies of the state, one stepped through the captured graph
    wa, _ = _full_window(torch.float32, window, seed=1)
    wb, _ = _full_window(torch.float32, window, seed=1)
    assert wa.capture(warmup=2)                                  # runs 2 real iterations before recording
    for _ in range(2):
        wb.iterate()
    for _ in range(3):
        wa.step()
        wb.iterate()
    # order-independent (exact fixed-point) assembly: replay and eager launches give the same BITS, whatever order the
    # workgroups and the two stream branches finish in
    assert torch.equal(wa.kf_poses, wb.kf_poses) and torch.equal(wa.H, wb.H)


# ------------------------------------------------------------------------------------------------
# Ragged and degenerate shapes: batch_photo_cost against the oracle on seeded random inputs
def _random_ba_case(seed, B, b_pairs, n, m, H, W, dt, behind=None):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    yy, xx = torch.meshgrid(torch.linspace(0, 3.1, H, dtype=torch.float64), torch.linspace(0, 4.2, W, dtype=torch.float64), indexing="ij")
    imgs = []
    for k in range(B):
        I = 0.5 + 0.3 * torch.sin(2.0 * xx + 0.7 * k) * torch.cos(1.5 * yy - 0.3 * k) + 0.1 * torch.sin(5 * xx * yy / 4)
        gx = torch.zeros_like(I); gy = torch.zeros_like(I)
        gx[:, 1:-1] = 0.5 * (I[:, 2:] - I[:, :-2]); gy[1:-1] = 0.5 * (I[2:] - I[:-2])
        imgs.append(torch.stack((I, gx, gy)))
    iag = torch.stack(imgs)
    K = torch.tensor([[0.9 * W, 0, (W - 1) / 2], [0, 0.9 * W, (H - 1) / 2], [0, 0, 1]], dtype=torch.float64)
    poses = torch.eye(4, dtype=torch.float64).repeat(B, 1, 1)
    poses[:, :3, 3] = 0.02 * rnd(B, 3)
    u = torch.rand(B, n, generator=g, dtype=torch.float64) * (W - 1)
    v = torch.rand(B, n, generator=g, dtype=torch.float64) * (H - 1)
    z = 1.0 + 0.3 * torch.rand(B, n, generator=g, dtype=torch.float64)
    Pw = torch.stack(((u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z), -1)
    if behind is not None:
        Pw[behind] = Pw[behind] * torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)      # a whole keyframe behind every camera
    ref = torch.arange(b_pairs) % B
    tgt = (ref + 1 + (torch.arange(b_pairs) // B)) % B
    D = 8 * B + 3 * m * B
    kf_inds = torch.arange(8 * B).reshape(B, 8)
    lm = 8 * B + torch.arange(3 * m * B).reshape(B, 3 * m)
    c = lambda t: t.to(dt)
    return {"vals": c(torch.rand(B, n, 1, generator=g, dtype=torch.float64)), "aff": c(0.05 * rnd(B, 2, 1)), "Pwn": c(Pw), "poses": c(poses),
            "iag": c(iag), "dT": c(rnd(B, n, 3, 6)), "dz": c(0.3 * rnd(B, n, 3, m, 1)), "dzdP": c(rnd(B, 1, 1, 3)), "K": c(K),
            "kf_inds": kf_inds, "lm": lm, "ref": ref, "tgt": tgt, "D": D}


@pytest.mark.parametrize("case", [
    dict(B=2, b_pairs=1, n=1, m=4, H=9, W=11),                     # one pixel, one pair, smallest m
    dict(B=3, b_pairs=4, n=1003, m=12, H=37, W=53),                # nothing a multiple of 64 / 256
    dict(B=3, b_pairs=6, n=257, m=64, H=16, W=24),                 # full m, a target visited twice
    dict(B=3, b_pairs=4, n=300, m=8, H=20, W=31, behind=1),        # every pixel of keyframe 1 invalid in its pairs
])
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_batch_photo_cost_ragged_vs_oracle(case, dt):
    import como_amd.odom.backend.photo as photo
    from oracle import photo_ba
    case = dict(case)
    C = _random_ba_case(17, dt=dt, **case)
    r, t = C["ref"], C["tgt"]
    args = lambda f: (f(C["vals"][r]), f(C["aff"][r]), f(C["Pwn"][r]), f(C["poses"][t]), f(C["aff"][t]), f(C["iag"][t]), f(C["dT"][r]),
                      f(C["dz"][r]), f(C["dzdP"][r]), f(C["kf_inds"][r]), f(C["kf_inds"][t]), f(C["lm"][r]), f(C["K"]))
    D = C["D"]
    Ho, go = torch.zeros((D, D), dtype=dt), torch.zeros(D, dtype=dt)
    eo, aux_o = photo_ba.batch_photo_cost(*args(lambda x: x), Ho, go, return_aux=True)
    H, g = torch.zeros((D, D), dtype=dt, device=DEV), torch.zeros(D, dtype=dt, device=DEV)
    e = photo.batch_photo_cost(*args(dev), H, g)
    valid = photo.last_aux["valid"].cpu().bool()
    nvalid = int(aux_o["valid"].sum())
    report("ba_ragged", case=str(case), dtype=str(dt), nvalid=nvalid, mask_mismatch=int((valid != aux_o["valid"]).sum()),
           H_rel=rel_err(H, Ho) if nvalid else 0.0)
    assert torch.equal(valid, aux_o["valid"])                          # masks bit-exact, whatever the shape
    assert torch.isfinite(H).all() and torch.isfinite(g).all()
    if nvalid == 0:
        assert H.abs().max().item() == 0.0 and g.abs().max().item() == 0.0
    else:
        tol = 1e-9 if dt == torch.float64 else 3e-4
        assert rel_err(H, Ho) < tol and rel_err(g, go) < tol
        assert abs(float(e) - float(eo)) <= tol * abs(float(eo))


@pytest.mark.parametrize("N,H,W", [(1, 9, 11), (63, 20, 31), (1003, 37, 53), (4099, 48, 64)])
def test_tracking_iter_ragged_vs_oracle(N, H, W):
    """tracking_iter on odd pixel counts / image sizes against the oracle (float32): mask bit-exact, delta within 2e-4."""
    import como_amd.odom.frontend.photo_tracking as pt
    from oracle import tracking as otrk
    g = torch.Generator().manual_seed(N)
    yy, xx = torch.meshgrid(torch.linspace(0, 3.1, H), torch.linspace(0, 4.2, W), indexing="ij")
    img = (0.5 + 0.3 * torch.sin(2.0 * xx) * torch.cos(1.5 * yy) + 0.1 * torch.sin(5 * xx * yy / 4)).float()
    K = torch.tensor([[0.9 * W, 0, (W - 1) / 2], [0, 0.9 * W, (H - 1) / 2], [0, 0, 1]])
    u = torch.rand(N, generator=g) * (W + 4) - 2                     # some points project outside the image
    v = torch.rand(N, generator=g) * (H + 4) - 2
    z = 1.0 + 0.3 * torch.rand(N, generator=g)
    P = torch.stack(((u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z), -1)
    if N > 2:
        P[1, 2] = -P[1, 2]                                           # one point behind the camera
    vals = torch.rand(N, generator=g)
    J = torch.randn(N, 8, generator=g)
    T = torch.eye(4)
    T[:3, 3] = torch.tensor([0.01, -0.02, 0.005])
    o = otrk.tracking_iter(T, P, K, img, torch.tensor([0.02, -0.01]), vals, J.clone())
    if not bool(o["valid"].any()):
        pytest.skip("no valid pixel in this draw")
    out = pt.tracking_iter(dev(T[None]), dev(P[None]), dev(K), dev(img[None, None]), dev(torch.tensor([0.02, -0.01]).reshape(1, 2, 1)),
                           dev(vals[None, :, None]), dev(J[None, :, None, :].clone()), 0.1, None)
    Tn, an, delta, mse, gn, pj, valid, depth = out
    assert torch.equal(valid[0].cpu(), o["valid"])
    nv = int(o["valid"].sum())
    if nv >= 8 and torch.isfinite(o["delta"]).all():                 # fewer valid rows than unknowns: singular in the reference too
        assert rel_err(delta[0, :, 0], o["delta"]) < 2e-3
        assert abs(float(mse) - float(o["mse"])) <= 1e-4 * abs(float(o["mse"])) + 1e-7


def test_blur_and_depth_pyramid_vs_torch():
    """GaussianBlurModule / DepthPyramidModule (pyr_depth, every mode) against the torch formulas of the reference."""
    import torch.nn.functional as F
    from como_amd.utils import image_processing as ip
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 1, 37, 52, generator=g, dtype=torch.float64)
    k = (1.0 / 16.0) * torch.tensor([[1.0, 2.0, 1.0], [2.0, 4.0, 2.0], [1.0, 2.0, 1.0]], dtype=torch.float64).view(1, 1, 3, 3)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), k)
    assert rel_err(ip.GaussianBlurModule(1, DEV, torch.float64)(dev(x)), ref) < 1e-14
    assert rel_err(ip.blur(dev(x.float())), ref) < 2e-6
    d = torch.rand(1, 1, 48, 64, generator=g) * 3 + 0.5
    d[0, 0, 10:20, 5:9] = float("nan")
    d[0, 0, 30:32, 40:42] = float("nan")                         # a fully-NaN 2x2 cell
    refs = {"nearest_neighbor": d[:, :, 0::2, 0::2], "max": F.max_pool2d(d, 2), "min": -F.max_pool2d(-d, 2)}
    mask = ~d.isnan()
    dm = torch.where(mask, d, torch.zeros_like(d))
    s_ = F.avg_pool2d(dm, 2, 2, divisor_override=1)
    c_ = F.avg_pool2d(mask.float(), 2, 2, divisor_override=1)
    refs["masked_bilinear"] = torch.where(c_ > 0, s_ / c_, torch.tensor(0.0))
    clean = torch.nan_to_num(d, nan=1.0)
    for mode, ref in refs.items():
        got = ip.pyr_depth(dev(d), mode).cpu()
        both_nan = got.isnan() & ref.isnan()
        assert torch.equal(got.isnan(), ref.isnan()), mode
        assert ((got - ref).abs()[~both_nan].max() if (~both_nan).any() else 0) < 1e-6, mode
    assert rel_err(ip.pyr_depth(dev(clean), "bilinear"), F.avg_pool2d(clean, 2, 2)) < 1e-6
    pyr = ip.DepthPyramidModule(0, 3, "nearest_neighbor", DEV)(dev(clean))
    assert [tuple(p.shape[-2:]) for p in pyr] == [(12, 16), (24, 32), (48, 64)]


def test_covariance_modules_vs_golden():
    """model.cov_modules / cross_cov_modules / diagonal_cov_modules (the objects Mapping.prep_predictor indexes) against the
    reference modules' float64 outputs."""
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.synth import depthcov_state_dict
    C = load_golden("cov_ops_f32.npz")
    model = DepthCovModule({k: dev(v) for k, v in depthcov_state_dict(0).items()})
    x1, E1, x2, E2 = (dev(C[k].double()) for k in ("x1", "E1", "x2", "E2"))
    K11 = model.cov_modules[-1](x1, E1)
    K12 = model.cross_cov_modules[-1](x1, E1, x2, E2)
    assert rel_err(K11, C["K11_py64"]) < 1e-7 and rel_err(K12, C["K12_py64"]) < 1e-7   # the twin casts coordinate differences to float32
    kd = model.diagonal_cov_modules[-1](x1, E1)
    assert rel_err(kd, torch.diagonal(C["K11_py64"], dim1=-2, dim2=-1)) < 1e-6      # k(x,x): Q = 0, safe_sqrt's 1e-8


def test_distill_depth_vs_golden():
    """distill_depth.py mirror (kernel matrices from the HIP covariance modules) against the reference's outputs, float32."""
    from como_amd.depth_cov.core import distill_depth as dd
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.synth import depthcov_state_dict
    G = load_golden("distill_f32.npz")
    model = DepthCovModule({k: dev(v) for k, v in depthcov_state_dict(0).items()})     # scales default to 1 as in ref_model()
    cov, cm, cn, z = dev(G["cov"]), dev(G["coords_m"]), dev(G["coords_n"]), dev(G["z_obs"])
    K_mm, K_nm, K_d = dd.calc_kernel_matrices(cm, cn, cov, model)
    assert rel_err(K_mm, G["K_mm"]) < 2e-6 and rel_err(K_nm, G["K_nm"]) < 2e-6 and rel_err(K_d, G["K_nn_diag"]) < 2e-6
    Kt, L_mm, sinv = dd.get_predictor(K_mm, K_nm, K_d)
    assert rel_err(L_mm, G["L_mm"]) < 1e-4 and rel_err(Kt, G["Kt"]) < 2e-3          # K_mm^-1 in float32: conditioning
    for wp in (0, 1):
        lz, res = dd.distill_depth_from_scratch(cm, cn, z, cov, model, bool(wp), 0.1)
        e = (lz.cpu() - G[f"logz_m_prior{wp}"]).abs().max().item()
        report("distill", prior=wp, logz_abs_err=e)
        assert e < (5e-3 if wp else 5e-2)                                           # un-regularised fit is ill-conditioned
        assert tuple(res.shape) == tuple(G[f"resid_prior{wp}"].shape)
    z1 = torch.exp(dev(G["logz_m_prior1"])[:, :5, :])
    lz2 = dd.distill_conditional_depth_from_scratch(cm, z1, cn, cov, z, model, 0.1, 0.05)
    assert (lz2.cpu() - G["logz_m2_cond"]).abs().max().item() < 5e-3


def test_track_and_init_vs_golden():
    """corr.py mirror: correspondences, newly sampled inducing points and their depths for a new keyframe, against the
    reference's track_and_init on the same inputs (mapping dtype float64).  The correspondence mask and the sampled pixel
    set must be identical; depths agree to the conditioning of the m x m float64 solves."""
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.odom.frontend.corr import track_and_init
    from como_amd.synth import depthcov_state_dict
    G = load_golden("corr_f64.npz")
    model = DepthCovModule({k: dev(v) for k, v in depthcov_state_dict(0).items()})
    corr_params = {"corr_mode": "logz", "corr_thresh": 3.0e-2, "distill_with_prior": True, "min_obs_depth": 0.0,
                   "logz_grad_mag_thresh": 7.0e-2}
    sampling = {"mode": "greedy_conditional_entropy", "max_num_coords": int(G["nmax"]), "max_stdev_thresh": 1.0e-2,
                "border": 3, "fixed_var": 0.0, "dist_thresh": 1.0e-1}
    H, W = G["z_img1"].shape[-2:]
    c2, z2, mask, call, zall = track_and_init(dev(G["pose1"]), dev(G["pose2"]), dev(G["coords_m1"]), dev(G["z_m1"]),
                                              dev(G["z_img1"]), dev(G["cov2"]), dev(G["K"]), model, corr_params, sampling,
                                              (H, W))
    assert torch.equal(mask.cpu(), G["corr_mask"])
    assert tuple(c2.shape) == tuple(G["coords_2"].shape) and torch.equal(c2.cpu().to(G["coords_2"].dtype), G["coords_2"])
    assert (call.cpu() - G["coords_all"]).abs().max().item() < 1e-9
    ez2 = (z2.cpu() - G["z2"]).abs().max().item()
    ezall = (zall.cpu() - G["z_all"]).abs().max().item()
    report("track_and_init", kept=int(mask.sum()), new=int(c2.shape[1]), z2_abs_err=ez2, z_all_abs_err=ezall)
    assert ezall < 1e-4 and ez2 < 1e-4


def test_tracking_state_machine_vs_golden():
    """Tracking mirror (como_amd/odom/Tracking.py) driven through the reference's sequence: keyframe reference set-up,
    six tracked frames, the keyframe / one-way requests they raise, and the re-basing on a second keyframe.  The decisions
    must be identical; poses agree to the float32 tracking tolerance."""
    from como_amd.odom.Tracking import Tracking
    G = load_golden("tracker_f32.npz")
    cfg = {"device": DEV, "dtype": "float", "color": "gray",
           "pyr": {"start_level": 0, "end_level": 3, "depth_interp_mode": "nearest_neighbor"},
           "term_criteria": {"max_iter": 50, "delta_norm": 1.0e-3, "rel_tol": 1.0e-3, "grad_norm": 1.0},
           "sigmas": {"photo": 1.0e-1},
           "keyframing": {"kf_depth_motion_ratio": 0.03, "kf_num_pixels_frac": 0.75, "one_way_freq": 3}}
    rgb, depth, T = dev(G["rgb"]), dev(G["depth"]), dev(G["poses_gt"])
    H, W = rgb.shape[-2:]
    trk = Tracking(cfg, G["K"].clone(), (H, W))
    trk.setup()
    aff0 = torch.zeros((1, 2, 1), device=DEV, dtype=torch.float32)
    trk.update_kf_reference(([1.0], rgb[0:1], T[0:1], aff0, depth[0:1]))
    kinds, second = [], int(G["second_kf_at"])
    worst_T = worst_med = 0.0
    worst_n = 0
    for k in range(1, rgb.shape[0]):
        if k - 1 == second:
            trk.update_kf_reference(([1.0 + second], rgb[second:second + 1], T[second:second + 1], aff0, depth[second:second + 1]))
            assert (trk.T_curr_kf.cpu() - G["rebased_T_curr_kf"]).abs().max().item() < 5e-6
            assert (trk.aff_curr_kf.cpu() - G["rebased_aff_curr_kf"]).abs().max().item() < 5e-6
        viz, mp = trk.handle_frame((1.0 + k, rgb[k:k + 1]))
        kinds.append(0 if mp is None else (1 if mp[0] == "keyframe" else 2))
        n_seen, med = trk.last_reproj_stats
        worst_T = max(worst_T, (trk.T_curr_kf.cpu() - G[f"T_curr_kf_{k}"]).abs().max().item(),
                      (viz[1].cpu() - G[f"T_w_curr_{k}"]).abs().max().item())
        worst_n = max(worst_n, abs(int(n_seen) - int(G[f"n_reproj_{k}"])))
        worst_med = max(worst_med, abs(float(med) - float(G[f"median_depth_{k}"])))
        assert (trk.aff_curr_kf.cpu() - G[f"aff_curr_kf_{k}"]).abs().max().item() < 5e-5
    report("tracker", kinds=kinds, pose_abs_err=worst_T, n_reproj_diff=worst_n, median_depth_err=worst_med)
    assert kinds == [int(x) for x in G["kinds"]]
    assert worst_T < 5e-6 and worst_n <= 3 and worst_med < 1e-5


@pytest.mark.parametrize("name,full,fused", [("ba_window_recent_f64.npz", False, True), ("ba_window_recent_f64.npz", False, False),
                                             ("ba_window_recent_full_f64.npz", True, True)])
def test_window_iterate_with_one_way_frames_vs_golden(name, full, fused):
    """Mapping.iterate()-equivalent with one-way (recent) frames in the window: before the window is full (mean-log-depth
    scale prior on keyframe 0: the fused HIP chain -- round 3: win_priors carries that prior -- and the reference-signature
    mirror path) and with a full window (landmark anchors; fused chain).  float64 pixel path."""
    import copy
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    G = load_golden(name)
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 2
    st = {k: dev(G[k]) for k in ("intrinsics", "kf_poses", "kf_aff_params", "kf_img_and_grads", "coords_m", "correspondence_mask",
                                 "P_m", "kf_timestamps", "obs_ref_mask", "pm_first_obs", "L_mm", "K_mm_inv", "pose_anchor",
                                 "recent_poses", "recent_aff_params", "recent_img_and_grads", "recent_timestamps")}
    st["Knm_Kmminv"] = prep_predictor(dev(G["cov_params_img"]), dev(G["coords_m"]), 1.0)[2]
    if full:
        st["P_anchor"] = dev(G["P_anchor"])
    else:
        st["init_scale_anchor"] = dev(G["init_scale_anchor"])
    wb = WindowBA(st, cfg=cfg, pix_dtype=torch.float64, window_full=full, fused=fused)
    assert wb.fused == fused and wb.F == wb.B + G["recent_poses"].shape[0]
    wb.median_depths.copy_(dev(G["median_depths_in"]))
    delta = wb.iterate()
    perr = (wb.kf_poses.cpu() - G["kf_poses_new"]).abs().max().item()
    rerr = (wb.recent_poses.cpu() - G["recent_poses_new"]).abs().max().item()
    aerr = (wb.recent_aff_params.cpu().reshape(-1) - G["recent_aff_new"].reshape(-1)).abs().max().item()
    report("window_iterate_recent", full=full, fused=fused, H_scaled=scaled_err(wb.H, G["H_full"]), delta_rel=rel_err(delta, G["delta"]), kf_pose_err=perr,
           recent_pose_err=rerr, recent_aff_err=aerr)
    assert scaled_err(wb.H, G["H_full"]) < 1e-8 and rel_err(wb.g, G["g_full"]) < 1e-8
    assert perr < 1e-9 and rerr < 1e-9 and aerr < 1e-9
    assert (wb.P_m.cpu() - G["P_new"]).abs().max().item() < 1e-8


@pytest.mark.parametrize("pix", ["double", "float"])
def test_mapping_state_machine_vs_golden(pix):
    """Mapping mirror (como_amd/odom/Mapping.py) driven through the reference's sequence: first keyframe, second keyframe,
    one-way frame, third keyframe, and a fourth that makes the 3-keyframe window slide, two GN iterations after every
    insertion (float64 pixel path).  Window bookkeeping (correspondence mask, first-observation masks, timestamps,
    landmark counts) must be identical; poses, landmarks and log-depths agree to what the float32 DepthCov network leaves
    (its covariance image feeds the sampler and the depth distillation of every new keyframe)."""
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.odom.Mapping import Mapping
    from como_amd.synth import depthcov_state_dict
    G = load_golden("mapping_f64.npz")
    cfg = {"device": DEV, "dtype": "double", "pix_dtype": pix, "color": "gray", "track_ref": {"num_keyframes": 1},
           "graph": {"num_keyframes": 3, "num_one_way_frames": 4}, "network_size": [32, 64], "graph_network": False,
           "photo_construction": {"nonmax_suppression_window": 2, "pairwise_batch_size": 128, "radius_thresh": 0.0,
                                  "degrees_thresh": 0.0},
           "sigmas": {"photo": 1.0e-1, "mean_depth_prior": 1.0e-2, "scale_prior": 1.0e-4, "pose_prior": 1.0e-6},
           "sampling": {"mode": "greedy_conditional_entropy", "max_num_coords": 12, "max_stdev_thresh": 1.0e-2, "border": 3,
                        "fixed_var": 0.0, "dist_thresh": 1.0e-1},
           "corr": {"corr_mode": "logz", "corr_thresh": 3.0e-2, "distill_with_prior": True, "min_obs_depth": 0.0,
                    "logz_grad_mag_thresh": 7.0e-2}}
    # "float": the production configuration (float32 per-pixel kernels, float64 system) -- same decisions, looser values
    loose = pix == "float"
    mp = Mapping(cfg, G["K"].clone())
    mp.setup(DepthCovModule({k: dev(v) for k, v in depthcov_state_dict(0).items()}))
    rgb = dev(G["rgb"])
    tags = [str(t) for t in G["snap_tags"]]
    worst = {"pose": 0.0, "P": 0.0, "logz": 0.0, "aff": 0.0, "med": 0.0}

    def check(i):
        g = lambda n: G[f"s{i}_{n}"]
        assert torch.equal(mp.correspondence_mask.cpu(), g("correspondence_mask")), tags[i]
        assert torch.equal(mp.obs_ref_mask.cpu(), g("obs_ref_mask")), tags[i]
        assert [float(t) for t in mp.kf_timestamps] == g("kf_timestamps").tolist(), tags[i]
        assert [float(t) for t in mp.recent_timestamps] == g("recent_timestamps").tolist(), tags[i]
        assert list(mp.depth_dims) == g("depth_dims").tolist() and bool(mp.window_full) == bool(g("window_full")), tags[i]
        want = g("pm_first_obs")
        got = mp.pm_first_obs.cpu()
        sampled = (want == want.round()).all(dim=-1)                      # freshly sampled points sit on integer pixels
        assert torch.equal(got[sampled], want[sampled]), tags[i]          # same sampled pixels
        assert (got - want).abs().max().item() < (2e-2 if loose else 1e-4), tags[i]   # tracked: reprojections through estimated poses
        worst["pose"] = max(worst["pose"], (mp.kf_poses.cpu() - g("kf_poses")).abs().max().item())
        worst["aff"] = max(worst["aff"], (mp.kf_aff_params.cpu() - g("kf_aff_params")).abs().max().item())
        worst["P"] = max(worst["P"], (mp.P_m.cpu() - g("P_m")).abs().max().item())
        worst["logz"] = max(worst["logz"], (mp.logzm.cpu() - g("logzm")).abs().max().item())
        worst["med"] = max(worst["med"], (mp.median_depths.cpu() - g("median_depths")).abs().max().item())
        if g("recent_poses").numel():
            worst["pose"] = max(worst["pose"], (mp.recent_poses.cpu() - g("recent_poses")).abs().max().item())
        assert (mp.pose_anchor.cpu() - g("pose_anchor")).abs().max().item() < (1e-3 if loose else 1e-4), tags[i]
        if f"s{i}_P_m_anchors" in G:
            assert (mp.P_m_anchors.cpu() - G[f"s{i}_P_m_anchors"]).abs().max().item() < (1e-2 if loose else 1e-3), tags[i]

    T0 = dev(G["poses_gt"])[0:1].clone()
    mp.init_keyframe(rgb[0:1], dev(G["cov_net0"]), dev(G["coords_m0"]), T0, dev(G["logz_m0"]),
                     torch.zeros((1, 2, 1), device=DEV, dtype=torch.float64), 1.0)
    mp.init_scale_anchor = dev(G["init_scale_anchor"])
    check(0)
    i = 1
    for k, ts, kind in ((1, 2.0, "kf"), (2, 2.5, "ow"), (3, 3.0, "kf"), (4, 4.0, "kf")):
        Tin, ain = dev(G[f"pose_init_{k}"]).clone(), dev(G[f"aff_init_{k}"]).clone()
        if kind == "kf":
            mp.add_keyframe(rgb[k:k + 1], Tin, ain, ts)
        else:
            mp.add_one_way_frame(rgb[k:k + 1], Tin, ain, ts)
        check(i)
        mp.iterate()
        mp.iterate()
        check(i + 1)
        i += 2
    report("mapping", pix=pix, snapshots=len(tags), landmarks=int(mp.P_m.shape[0]), **{k + "_abs_err": v for k, v in worst.items()})
    if loose:
        assert worst["pose"] < 1e-4 and worst["aff"] < 1e-3 and worst["P"] < 1e-2 and worst["logz"] < 1e-2 and worst["med"] < 5e-3
    else:
        # median_depths is the reference's store_vars value (full depth image) at every snapshot
        assert worst["pose"] < 1e-5 and worst["aff"] < 1e-4 and worst["P"] < 1e-3 and worst["logz"] < 1e-3 and worst["med"] < 5e-4


def test_two_frame_init_state_machine_vs_golden():
    """TwoFrameSfm.handle_frame and Mapping.attempt_two_frame_init mirrors against the reference on the same frames: same
    frame at which initialisation succeeds, same inducing pixels, relative pose / log-depth codes of the SfM within the
    tolerance left by the float32 network, and the two keyframes created from it."""
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.odom.frontend.TwoFrameSfm import TwoFrameSfm
    from como_amd.odom.Mapping import Mapping
    from como_amd.synth import depthcov_state_dict
    G = load_golden("sfm_init_f64.npz")
    cfg = {"device": DEV, "dtype": "double", "pix_dtype": "double", "color": "gray", "track_ref": {"num_keyframes": 1},
           "graph": {"num_keyframes": 3, "num_one_way_frames": 4}, "network_size": [32, 64], "graph_network": False,
           "photo_construction": {"nonmax_suppression_window": 2, "pairwise_batch_size": 128, "radius_thresh": 0.0,
                                  "degrees_thresh": 0.0},
           "term_criteria": {"max_iter": 20, "delta_norm": 1.0e-8, "abs_tol": 1.0e-6, "rel_tol": 1.0e-6},
           "sigmas": {"photo": 1.0e-1, "mean_depth_prior": 1.0e-2, "scale_prior": 1.0e-4, "pose_prior": 1.0e-6},
           "sampling": {"mode": "greedy_conditional_entropy", "max_num_coords": 12, "max_stdev_thresh": 1.0e-2, "border": 3,
                        "fixed_var": 0.0, "dist_thresh": 1.0e-1},
           "corr": {"corr_mode": "logz", "corr_thresh": 3.0e-2, "distill_with_prior": True, "min_obs_depth": 0.0,
                    "logz_grad_mag_thresh": 7.0e-2},
           "init": {"start_level": 0, "end_level": 3, "max_iter": 50, "delta_norm": 1.0e-4, "rel_tol": 1.0e-4,
                    "kf_depth_motion_ratio": 0.04, "kf_num_pixels_frac": 0.75}}
    model = DepthCovModule({k: dev(v) for k, v in depthcov_state_dict(0).items()})
    rgb = dev(G["rgb"])
    K = dev(G["K"])
    sfm = TwoFrameSfm(cfg, K, model, -1, [32, 64])
    flags = []
    worst_T = worst_d = 0.0
    for k in range(len(G["is_init_flags"])):
        r = sfm.handle_frame(rgb[k:k + 1], 1.0 + k)
        flags.append(bool(r[0]))
        if k == 0:
            assert torch.equal(sfm.coords_m.cpu(), G["coords_m"])                      # same inducing pixels
            assert rel_err(sfm.cov_params_img, G["cov_params_img"]) < 2e-4              # float32 network, 30 layers
        else:
            worst_T = max(worst_T, (r[1].cpu() - G[f"T_curr_kf_{k}"]).abs().max().item())
            worst_d = max(worst_d, (r[3].cpu() - G[f"logd_{k}"]).abs().max().item(),
                          (r[6].cpu() - G[f"mean_log_depth_{k}"]).abs().max().item())
    assert flags == [bool(x) for x in G["is_init_flags"]]
    mp = Mapping(cfg, G["K"].clone())
    mp.setup(model)
    done = -1
    for k in range(rgb.shape[0]):
        if mp.attempt_two_frame_init(1.0 + k, rgb[k:k + 1]):
            done = k
            break
    assert done == int(G["init_done_at"]) and [float(t) for t in mp.kf_timestamps] == G["m_kf_timestamps"].tolist()
    assert torch.equal(mp.correspondence_mask.cpu(), G["m_correspondence_mask"])
    assert torch.equal(mp.obs_ref_mask.cpu(), G["m_obs_ref_mask"])
    e_pose = (mp.kf_poses.cpu() - G["m_kf_poses"]).abs().max().item()
    e_P = (mp.P_m.cpu() - G["m_P_m"]).abs().max().item()
    e_lz = (mp.logzm.cpu() - G["m_logzm"]).abs().max().item()
    e_anchor = (mp.init_scale_anchor.cpu() - G["m_init_scale_anchor"]).abs().max().item()
    report("two_frame_init", init_at=done, sfm_pose_err=worst_T, sfm_logd_err=worst_d, kf_pose_err=e_pose, P_err=e_P, logz_err=e_lz,
           scale_anchor_err=e_anchor)
    assert worst_T < 2e-5 and worst_d < 2e-3
    assert e_pose < 2e-5 and e_P < 2e-3 and e_lz < 2e-3 and e_anchor < 2e-4


def test_sequential_odometry_vs_golden():
    """The whole headless odometry loop (como_amd/odom/sequential.py: two-frame initialisation, float32 tracking, keyframe
    / one-way requests, one float64 mapping iteration per frame, tracker reference refresh) against the reference's
    sequential mode on the same 16 rendered frames: the same request for every frame, the same keyframe / one-way
    timestamps and correspondence mask at the end, tracked world poses within 1e-4."""
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
    from como_amd.odom.sequential import ComoSeq
    from como_amd.synth import depthcov_state_dict
    G = load_golden("odometry_seq.npz")
    mcfg = {"device": DEV, "dtype": "double", "pix_dtype": "double", "color": "gray", "track_ref": {"num_keyframes": 1},
            "graph": {"num_keyframes": 3, "num_one_way_frames": 4}, "network_size": [32, 64], "graph_network": False,
            "photo_construction": {"nonmax_suppression_window": 2, "pairwise_batch_size": 128, "radius_thresh": 0.0,
                                   "degrees_thresh": 0.0},
            "term_criteria": {"max_iter": 20, "delta_norm": 1.0e-8, "abs_tol": 1.0e-6, "rel_tol": 1.0e-6},
            "sigmas": {"photo": 1.0e-1, "mean_depth_prior": 1.0e-2, "scale_prior": 1.0e-4, "pose_prior": 1.0e-6},
            "sampling": {"mode": "greedy_conditional_entropy", "max_num_coords": 12, "max_stdev_thresh": 1.0e-2, "border": 3,
                         "fixed_var": 0.0, "dist_thresh": 1.0e-1},
            "corr": {"corr_mode": "logz", "corr_thresh": 3.0e-2, "distill_with_prior": True, "min_obs_depth": 0.0,
                     "logz_grad_mag_thresh": 7.0e-2},
            "init": {"start_level": 0, "end_level": 3, "max_iter": 50, "delta_norm": 1.0e-4, "rel_tol": 1.0e-4,
                     "kf_depth_motion_ratio": 0.04, "kf_num_pixels_frac": 0.75}}
    tcfg = {"device": DEV, "dtype": "float", "color": "gray",
            "pyr": {"start_level": 0, "end_level": 3, "depth_interp_mode": "nearest_neighbor"},
            "term_criteria": {"max_iter": 50, "delta_norm": 1.0e-3, "rel_tol": 1.0e-3, "grad_norm": 1.0},
            "sigmas": {"photo": 1.0e-1},
            "keyframing": {"kf_depth_motion_ratio": 0.05, "kf_num_pixels_frac": 0.75, "one_way_freq": 3}}
    model = DepthCovModule({k: dev(v) for k, v in depthcov_state_dict(0).items()})
    rgb = dev(G["rgb"])
    H, W = rgb.shape[-2:]
    odo = ComoSeq({"tracking": tcfg, "mapping": mcfg}, G["K"].clone(), (H, W), model)
    code = {None: 0, "init": 3, "keyframe": 1, "one-way": 2}
    kinds, worst = [], 0.0
    for k in range(rgb.shape[0]):
        n_before = len(odo.est_poses)
        kinds.append(code[odo.iter(1.0 + k, rgb[k:k + 1])])
        if len(odo.est_poses) > n_before:
            worst = max(worst, (odo.est_poses[-1].cpu().double() - G[f"T_w_curr_{k}"].double()).abs().max().item())
        nk = odo.mapping.kf_poses.shape[0] if odo.mapping.kf_poses.dim() > 1 else 0
        assert nk == int(G[f"n_kf_{k}"]), k
    mp = odo.mapping
    report("odometry_seq", kinds=kinds, tracked_pose_abs_err=worst, kf_pose_err=(mp.kf_poses.cpu() - G["m_kf_poses"]).abs().max().item(),
           P_err=(mp.P_m.cpu() - G["m_P_m"]).abs().max().item())
    assert kinds == [int(x) for x in G["kinds"]]
    assert [float(t) for t in mp.kf_timestamps] == G["m_kf_timestamps"].tolist()
    assert [float(t) for t in mp.recent_timestamps] == G["m_recent_timestamps"].tolist()
    assert torch.equal(mp.correspondence_mask.cpu(), G["m_correspondence_mask"])
    assert worst < 1e-4 and (mp.kf_poses.cpu() - G["m_kf_poses"]).abs().max().item() < 1e-4
