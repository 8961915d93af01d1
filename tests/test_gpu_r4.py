"""Round-4 pins of the HIP path (through the C ABI), `-m gpu` on an MI355X:
  * the float64 select with ONE candidate exchange for digits 3..5 (csrc/select.hip como_select_cand_*): ranks emulated as slices
    of one array, the histogram all-reduces as sums, the all-gather as a concatenation -- against torch.median, with ties, empty
    ranks, several segments, and the not-representable case (more than 512 candidates on a rank);
  * own conditioning (K_mm^-1 from csrc/smallsolve.hip) in the remaining full-size reference pins (float32, window 1, config 4)."""
import numpy as np
import pytest
import torch

from tests.conftest import report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _exchange_median(x, valid, world):
    """x (nseg, n) float64 on the device; `world` ranks own contiguous column ranges.  Returns (nseg,) medians."""
    from como_amd import _lib
    L = _lib.lib()
    nseg, n = x.shape
    s = _lib.stream_ptr(x.device)
    words = L.como_select_workspace_bytes() // 4
    cw = L.como_select_cand_words()
    per = (n + world - 1) // world
    sl = [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]
    xs = [x[:, b:e].contiguous() for b, e in sl]
    vs = [None if valid is None else valid[:, b:e].to(torch.uint8).contiguous() for b, e in sl]
    hs = [torch.zeros(nseg * words, dtype=torch.int32, device=x.device) for _ in range(world)]
    for p in range(3):
        for r in range(world):
            if xs[r].shape[1] > 0:
                _lib.check(L.como_select_hist_f64(xs[r].data_ptr(), _lib.ptr(vs[r]), xs[r].shape[1], nseg, hs[r].data_ptr(), p, s), "hist")
        tot = sum(h.view(nseg, 6, 2048)[:, p] for h in hs)                 # the all-reduce of digit p
        for h in hs:
            h.view(nseg, 6, 2048)[:, p] = tot
    loc = torch.zeros((world, nseg, cw), dtype=torch.int32, device=x.device)
    for r in range(world):
        if xs[r].shape[1] > 0:
            _lib.check(L.como_select_hist_f64(xs[r].data_ptr(), _lib.ptr(vs[r]), xs[r].shape[1], nseg, hs[r].data_ptr(), 3 | 0x300, s), "collect")
        _lib.check(L.como_select_cand_pack(hs[r].data_ptr(), nseg, loc[r].data_ptr(), s), "pack")
    outs = []
    for r in range(world):                                                  # every rank merges the same gathered records
        _lib.check(L.como_select_cand_merge(hs[r].data_ptr(), nseg, loc.data_ptr(), world, nseg, 0, s), "merge")
        o = torch.empty((nseg, 3), dtype=torch.float64, device=x.device)
        _lib.check(L.como_select_finish_f64(hs[r].data_ptr(), nseg, o.data_ptr(), s), "finish")
        outs.append(o)
    for o in outs[1:]:
        assert torch.equal(torch.nan_to_num(o, nan=-1.0), torch.nan_to_num(outs[0], nan=-1.0))   # every rank: the same bits
    return outs[0][:, 0], outs[0][:, 2], loc[:, :, 0]


@pytest.mark.parametrize("world", [1, 2, 8])
def test_double_select_candidate_exchange(world):
    g = torch.Generator().manual_seed(17 + world)
    nseg, n = 3, 200_003
    x = (torch.randn((nseg, n), generator=g, dtype=torch.float64) * torch.exp(2 * torch.randn((nseg, n), generator=g, dtype=torch.float64))).abs()
    x[:, ::7] = 0.125                                           # ties away from the median
    x[1, ::3] = x[1].median()                                   # and a third of segment 1 tied AT its median (identical keys)
    valid = torch.rand((nseg, n), generator=g) < 0.8
    valid[2, n // 3:] = False                                   # the last ranks hold no valid key of segment 2
    xd, vd = x.to(DEV), valid.to(DEV)
    med, nv, cnt = _exchange_median(xd, vd, world)
    for sgm in range(nseg):
        ref = torch.median(x[sgm][valid[sgm]])
        if sgm == 1:                                            # ~53 k identical keys: more than 512 candidates on a rank
            # round 5: an overflowing candidate list of ONE value travels as (count, key) and the median is exact; only if a rank's
            # overflowing list also holds another key with the same 33 leading bits it is unrepresentable -- then it is REPORTED
            # (NaN, zero valid keys), never silently wrong
            assert int(cnt[:, 1].max()) > 512
            if torch.isnan(med[1]):
                assert nv[1] == 0
            else:
                assert med[1].item() == ref.item() and int(nv[1]) == int(valid[1].sum())
        else:
            assert med[sgm].item() == ref.item() and int(nv[sgm]) == int(valid[sgm].sum())
            assert int(cnt[:, sgm].max()) <= 512
    report("select_exchange", world=world, cand_max=[int(c) for c in cnt.max(0).values])
    # no mask, one segment, the plain six-pass select as the second opinion
    from como_amd.utils.select import masked_median
    y = xd[0:1].contiguous()
    m2, _, _ = _exchange_median(y, None, world)
    assert m2[0].item() == masked_median(y)[0].item() == torch.median(x[0]).item()


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fixture,pix,window", [("fullsize_window4.npz", torch.float32, 4), ("fullsize_window1.npz", torch.float64, 1),
                                                ("fullsize_window1.npz", torch.float32, 1), ("fullsize_window32.npz", torch.float64, 4)])
def test_fullsize_windows_with_own_conditioning(fixture, pix, window):
    """The full-size reference pins with NOTHING of the reference's in the path (round 3 had this for window 4 / float64 only):
    K_mm + 1e-6 I factored and inverted by csrc/smallsolve.hip instead of the fixture's K_mm^-1 -- the float32 pixel path, the
    dense window (window 1 = the bench headline) and config 4 (32 keyframes, D = 2680).  cond(K_mm) ~ 1e8: two correct float64
    inverses give K~ that differ by ~1e-8 relative; the solved poses must stay within the bars of the pins that take the
    reference's inverse (1e-7 float64 / 1e-4 float32 -- the north star's 1e-4)."""
    from tests.conftest import load_golden, rel_err
    from tests.test_gpu_r2 import _window_from_seed
    import como_amd.odom.backend.linear_system as ls
    G = load_golden(fixture)
    f64 = pix == torch.float64
    wb, st = _window_from_seed(G, pix, window, own_inverse=True)
    e_inv = rel_err(st["K_mm_inv"], G["K_mm_inv"])
    iters = sum(1 for k in G if k.endswith("_delta"))
    worst = {"pose": 0.0, "aff": 0.0, "P": 0.0, "med": 0.0}
    for it in range(iters):
        gi = lambda k: G[f"it{it}_{k}"]
        wb.iterate()
        torch.cuda.synchronize()
        worst["pose"] = max(worst["pose"], (wb.kf_poses.cpu() - gi("kf_poses_new")).abs().max().item())
        worst["aff"] = max(worst["aff"], (wb.kf_aff_params.cpu() - gi("kf_aff_new")).abs().max().item())
        worst["P"] = max(worst["P"], (wb.P_m.cpu() - gi("P_new")).abs().max().item())
        worst["med"] = max(worst["med"], ((wb.median_depths.cpu() - gi("median_depths_full")).abs() / gi("median_depths_full")).max().item())
    report("fullsize_own_conditioning_r4", fixture=fixture, pix=str(pix), window=window, D=wb.dim, iters=iters, Kinv_rel_vs_reference=e_inv,
           info=int(ls.solve_system.last_info), **worst)
    assert int(ls.solve_system.last_info) == 0
    assert worst["pose"] < (1e-7 if f64 else 1e-4) and worst["aff"] < (1e-7 if f64 else 1e-4)
    assert worst["P"] < (1e-4 if f64 else 2e-3) and worst["med"] < (1e-6 if f64 else 1e-5)


# ---------------------------------------------------------------------------------------------------------------------
def test_weighted_normal_equations_kernel():
    """csrc/gram.hip (como_gram_f64) against torch: A^T W A, A^T W (y - A c), the statistics; masked rows hold NaN (they must
    contribute exact zeros); a row-strided view; sizes that are not multiples of the kernel's row steps; exactly symmetric."""
    from tests.conftest import rel_err
    from como_amd.depth_cov.core.distill_depth import gram_weighted
    g = torch.Generator().manual_seed(5)
    for n, m, strided in ((49_152, 64, False), (1_003, 64, True), (37, 16, False), (300_001, 48, False)):
        A = torch.randn((1, n, 64 if strided else m), generator=g, dtype=torch.float64)
        w = torch.rand((1, n, 1), generator=g, dtype=torch.float64) * (torch.rand((1, n, 1), generator=g) < 0.8)
        y = torch.randn((1, n, 1), generator=g, dtype=torch.float64)
        c = torch.randn((1, m, 1), generator=g, dtype=torch.float64)
        Ad = A.to(DEV)
        Av = Ad[:, :, :m] if strided else Ad
        dead = (w == 0)[0, :, 0]
        Ad[0, dead.to(DEV)] = float("nan")                      # masked rows may hold anything
        yd = y.clone()
        yd[0, dead] = float("nan")
        AtA, Atb, st = gram_weighted(Av, w.to(DEV), yd.to(DEV), c=c.to(DEV), want_stats=True)
        Ar = A[:, :, :m]
        r = y - Ar @ c
        ref_AtA = Ar.mT @ (w * Ar)
        ref_Atb = Ar.mT @ (w * r)
        assert rel_err(AtA, ref_AtA) < 1e-12 and rel_err(Atb, ref_Atb) < 1e-12, (n, m)
        assert torch.equal(AtA, AtA.mT)
        ref_st = torch.stack((w.sum(), (w * r).sum(), (w * r * r).sum(), (w != 0).sum().double()))
        assert rel_err(st, ref_st) < 1e-12
        AtA2, Atb2 = gram_weighted(Av.nan_to_num(0.0), None, y.to(DEV))            # no weights, no c
        assert rel_err(AtA2, torch.where(dead[None, :, None], torch.zeros_like(Ar), Ar).mT @ torch.where(dead[None, :, None], torch.zeros_like(Ar), Ar)) < 1e-12
    report("gram_kernel", ok=True)


@pytest.mark.gpu
def test_rebuilt_window_with_inherited_workspaces_equals_fresh_window():
    """The sequential loop rebuilds the window solver on every one-way frame and keyframe with `WindowBA(prev=...)`: topology
    tables (same keyframe set), scratch workspaces and -- with the same predictors -- the band-median state are taken over from the
    retired window.  None of it may change a result: after the same two iterations the rebuilt window equals a window built from
    scratch bit for bit (the chain is order-independent by construction), for a one-way frame added to the same keyframes AND for a
    window over other predictors that reuses the planes; and `snapshot_state` returns what the separate tensors hold."""
    import copy
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    pix = torch.float32

    def predictor(cov, cm):
        Kinv, L, Kt = prep_predictor(cov.double(), cm.double(), 1.0)
        return Kinv, L, Kt.to(pix)
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = 4
    st = synth.make_window(B=4, H=192, W=256, m=64, dtype=torch.float64, device=DEV, seed=5, predictor=predictor)
    st["Knm_Kmminv_pix"] = st["Knm_Kmminv"]                    # (already in the pixel type: every window aliases the same K~)
    w0 = WindowBA(st, cfg=cfg, pix_dtype=pix, window_full=True)
    for _ in range(3):
        w0.step()
    sn = w0.snapshot_state()
    assert torch.equal(sn["poses"], w0.poses_all) and torch.equal(sn["aff"], w0.aff_all) and torch.equal(sn["P_m"], w0.P_m)
    assert torch.equal(sn["median"], w0.median_depths) and torch.equal(sn["pm"], w0.w["pm"]) and torch.equal(sn["logzm"], w0.w["logzm"])
    assert sn["poses"].data_ptr() != w0.poses_all.data_ptr()

    def state_after(w):
        out = dict(st)
        out.update({"kf_poses": sn["poses"][:4].clone(), "kf_aff_params": sn["aff"][:4].reshape(4, 2, 1).clone(), "P_m": sn["P_m"].clone(),
                    "median_depth_init": sn["median"].clone()})
        return out
    # (1) same keyframes + a one-way frame (a copy of keyframe 3's image at a perturbed pose)
    s1 = state_after(w0)
    T = s1["kf_poses"][3:4].clone()
    T[0, 0, 3] += 2e-3
    s1.update({"recent_poses": T, "recent_aff_params": torch.zeros((1, 2, 1), dtype=torch.float64, device=DEV),
               "recent_img_and_grads": st["kf_img_and_grads"][3:4].clone(), "recent_timestamps": torch.tensor([3.5], dtype=torch.float64)})
    res = {}
    for how in ("prev", "fresh"):
        w = WindowBA(s1, cfg=cfg, pix_dtype=pix, window_full=True, prev=w0 if how == "prev" else None)
        if how == "prev":
            assert w.w["ba_ws"] is w0.w["ba_ws"] and len(w.w["dr_ws"]) > 0          # workspaces (and band state) inherited
            band = [e["band"] for k, e in w.w["dr_ws"].items() if k[0] == "full"][0]
            assert band["ident"] is not None and band["calls"] == 3
        for _ in range(2):
            w.step()
        res[how] = {k: v.clone() for k, v in w.snapshot_state().items()}
        w1 = w
    for k in res["fresh"]:
        assert torch.equal(res["prev"][k], res["fresh"][k]), k
    # (2) other predictors (another keyframe set): planes reused, band state rebuilt in them
    st2 = synth.make_window(B=4, H=192, W=256, m=64, dtype=torch.float64, device=DEV, seed=6, predictor=predictor)
    res = {}
    for how in ("prev", "fresh"):
        w = WindowBA(st2, cfg=cfg, pix_dtype=pix, window_full=True, prev=w1 if how == "prev" else None)
        if how == "prev":
            assert w.w["ba_ws"] is w1.w["ba_ws"]
        for _ in range(2):
            w.step()
        res[how] = {k: v.clone() for k, v in w.snapshot_state().items()}
    for k in res["fresh"]:
        assert torch.equal(res["prev"][k], res["fresh"][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_reproject_points_kernel_vs_torch_formulas(dt):
    """csrc/trackref.hip `como_reproject_points_*` against the torch expressions it replaces in the correspondence search
    (frontend/corr.py reproject_points + filter_reproj_coords = reference corr.py:17-43): camera points and row/col coordinates to
    rounding (the rigid transform accumulates in k order, the library product it replaces in its own), the in-image / depth test
    identical away from the boundaries, kept points in index order; explicit coordinates and the pixel-grid form."""
    from como_amd import synth
    from como_amd.odom.frontend import corr
    from como_amd.utils.coords import get_test_coords
    g = torch.Generator().manual_seed(11)
    H, W = 96, 128
    K = synth.intrinsics_for(H, W).to(dt)[None].to(DEV)
    T = synth.gt_poses(3, step=0.05, deg=2.0)[2:3].to(dt).to(DEV)
    z = (1.0 + torch.rand((1, H * W, 1), generator=g, dtype=torch.float64)).to(dt).to(DEV)
    z[0, ::17, 0] = 1e-4                                        # some points below the depth threshold
    grid = get_test_coords((H, W), device=DEV)
    tol = 1e-12 if dt == torch.float64 else 2e-5
    for coords in (None, (grid.to(dt) + 0.25)):
        c_in = grid if coords is None else coords
        rc_ref, P_ref = corr.reproject_points(c_in, z, T, K)
        rc_f, P_f, keep_ref = corr.filter_reproj_coords(rc_ref, P_ref, (H, W), 0.05)
        rc, P, keep = corr.reproject_and_filter(coords, z, T, K, (H, W), 0.05, grid_width=W if coords is None else None)
        near = ((rc_ref[0] - 1).abs().min(dim=1).values < 1e-6) | ((rc_ref[0, :, 0] - (H - 1)).abs() < 1e-6) | \
               ((rc_ref[0, :, 1] - (W - 1)).abs() < 1e-6)
        assert not bool(near.any())                            # (no point of this case sits on a boundary of the test)
        assert torch.equal(keep, keep_ref) and 0 < int(keep.sum()) < H * W
        assert rc.shape == rc_f.shape and P.shape == P_f.shape
        assert float((rc - rc_f).abs().max()) <= tol * W and float((P - P_f).abs().max()) <= tol * 4
        rc2, P2, none = corr.reproject_and_filter(coords if coords is not None else grid.to(dt), z, T, K)
        assert none is None and float((rc2 - rc_ref).abs().max()) <= tol * W * 40 and float((P2 - P_ref).abs().max()) <= tol * 4
    report("reproject_points_kernel", dtype=str(dt), kept=int(keep.sum()), of=H * W)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(4999, 64), (307200, 61), (37, 8), (16, 1)])
def test_predictor_kernel_vs_torch(n, m):
    """csrc/gram.hip `como_predictor_f64` through `distill_depth.get_predictor` against the torch expressions of the reference
    (distill_depth.py:30-48): K~ = K_nm K_mm^-1, the conditional variances and their inverse standard deviations; padded form
    (m not a multiple of 4): the extra columns are exact zeros and the view has the padded row stride `gram_weighted` reads."""
    from como_amd.depth_cov.core import distill_depth as dd
    from como_amd.utils.lin_alg import chol_small
    g = torch.Generator().manual_seed(n + m)
    X = torch.randn((1, m, m + 3), generator=g, dtype=torch.float64)
    K_mm = (X @ X.mT / (m + 3) + 0.5 * torch.eye(m, dtype=torch.float64)).to(DEV)
    K_nm = torch.randn((1, n, m), generator=g, dtype=torch.float64).to(DEV)
    diag = (10.0 + torch.rand((1, n), generator=g, dtype=torch.float64)).to(DEV) * m      # (> k^T K_mm^-1 k: positive variances)
    inv = chol_small(K_mm, want_L=True, want_inv=True)["inv"]
    Kt_ref = K_nm @ inv
    var_ref = diag - torch.sum(K_nm * Kt_ref, dim=2)
    var_ref = var_ref + (torch.min(var_ref) + 1e-8)
    sinv_ref = 1.0 / torch.sqrt(var_ref.unsqueeze(-1))
    for pad4 in (False, True):
        Kt, L_mm, sinv = dd.get_predictor(K_mm, K_nm, diag, pad4=pad4)
        scale = float(Kt_ref.abs().max())
        assert Kt.shape == (1, n, m) and float((Kt - Kt_ref).abs().max()) <= 1e-13 * scale * m
        assert float(((sinv - sinv_ref) / sinv_ref).abs().max()) <= 1e-9
        full = dd.padded_predictor(Kt)
        if pad4 and m % 4:
            assert full.shape[2] == (m + 3) // 4 * 4 and float(full[:, :, m:].abs().max()) == 0.0 and Kt.stride(1) == full.shape[2]
        else:
            assert full is Kt
    report("predictor_kernel", n=n, m=m, max_abs_err=float((Kt - Kt_ref).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_se3_compose_kernel_vs_torch(dt):
    """csrc/window.hip `como_se3_compose_*` (lie_algebra.composeSE3) against the torch forms of get_rel_pose / get_T_w_curr
    (reference transforms.py:6-13): inv(A) B and A inv(B) for batches and broadcast operands."""
    from como_amd import synth
    from como_amd.geometry.lie_algebra import composeSE3, invertSE3
    g = torch.Generator().manual_seed(4)
    A = synth.se3_exp(0.3 * torch.randn((7, 6), generator=g, dtype=torch.float64)).to(dt).to(DEV)
    B = synth.se3_exp(0.3 * torch.randn((7, 6), generator=g, dtype=torch.float64)).to(dt).to(DEV)
    tol = 1e-14 if dt == torch.float64 else 1e-6
    for a, b in ((A, B), (A[:1], B), (A, B[3:4])):
        for mode, ref in ((0, a @ b), (1, invertSE3(a) @ b), (2, a @ invertSE3(b))):
            out = composeSE3(a, b, mode)
            assert out.shape == ref.shape and float((out - ref).abs().max()) <= tol * 4
            assert torch.equal(out[:, 3], torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=dt, device=DEV).expand(out.shape[0], 4))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_rgb_to_gray_kernel_equals_torch_expression(dt):
    """csrc/image.hip `como_rgb_to_gray_*` = the five-launch torch expression of rgb_to_grayscale, bit for bit (same products and
    sums, each rounded on its own)."""
    from como_amd.utils.image_processing import rgb_to_grayscale
    g = torch.Generator().manual_seed(2)
    rgb = torch.rand((2, 3, 37, 53), generator=g, dtype=torch.float64).to(dt).to(DEV)
    r, gg, b = rgb.unbind(dim=-3)
    ref = (0.2989 * r + 0.587 * gg + 0.114 * b).unsqueeze(-3)
    out = rgb_to_grayscale(rgb)
    assert out.shape == ref.shape and torch.equal(out, ref)
    assert torch.equal(rgb_to_grayscale(rgb.cpu()), ref.cpu())
