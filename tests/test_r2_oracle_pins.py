"""CPU: the oracle pinned against round-2 reference fixtures (tests/golden/make_golden_r2.py -- every value there was
produced by the reference's own Mapping.iterate): landmark re-initialisation, the 32-keyframe window (config 4), the metric
configuration (8 keyframes, 640x480, m = 64, nonmax window 4) and the SE(3) exponential against scipy's expm."""
import pytest
import torch

from tests.conftest import load_golden, scaled_err
from oracle import depthcov, geom
from oracle.window import OracleWindow


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300)).item()


def test_se3_exp_vs_scipy_expm():
    """Independent pin of the SE(3) exponential: T = expm([[w]x, v], [0, 0]) (scipy) for COMO's xi = [omega, v], i.e. what
    lietorch's SE3.exp([tau = v, phi = omega]) documents and what the reference feeds it (lie_algebra.py:45-56).
    Checked: the oracle's closed form, the product's mirror, and the golden-generation shim (so the fixtures that went
    through the shim are consistent with expm too)."""
    import importlib.util
    import os
    from como_amd.geometry import lie_algebra as la
    G = load_golden("se3_expm.npz")
    xi, want = G["xi_omega_v"], G["expm"]
    assert (geom.se3_exp(xi) - want).abs().max() < 1e-12
    assert (la.se3_exp(xi) - want).abs().max() < 1e-12
    assert (la.batch_se3(G["T0"], xi) - G["T0_expm"]).abs().max() < 1e-12
    spec = importlib.util.spec_from_file_location("lietorch_shim", os.path.join(os.path.dirname(__file__), "golden", "_shims",
                                                                               "lietorch.py"))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    tau_phi = torch.cat((xi[:, 3:], xi[:, :3]), dim=1)            # the reordering of lie_algebra.py:45-56
    assert (shim.SE3.exp(tau_phi).matrix() - want).abs().max() < 1e-12
    # float32 (tracking dtype)
    assert (la.se3_exp(xi.float()).double() - want).abs().max() < 5e-6


def _state(G, keys):
    return {k: G[k] for k in keys}


def test_oracle_reinit_vs_reference():
    """Landmarks behind a camera / below 0.1 x median: scaffold outputs, the landmarks moved for good and the whole
    iteration against the reference (Mapping.py:603-659, sparse_map.py:26-41)."""
    G = load_golden("ba_window_reinit_f64.npz")
    st = _state(G, ("intrinsics", "kf_poses", "kf_aff_params", "kf_img_and_grads", "coords_m", "correspondence_mask", "P_m",
                    "kf_timestamps", "obs_ref_mask", "pm_first_obs", "L_mm", "K_mm_inv", "Knm_Kmminv", "pose_anchor", "P_anchor"))
    st["median_depth_init"] = G["median_depths_in"]
    ow = OracleWindow(st, window=2)
    delta = ow.iterate()
    assert int(ow.moved.sum()) == int(G["n_moved"]) >= 2
    assert int(ow.zmask.sum()) > int(ow.moved.sum())              # some replacements are per-frame only
    assert torch.equal(ow.moved, (G["sc_P_m_after"] != G["P_m"]).any(dim=1))
    assert rel(ow.H, G["it0_H_full"]) < 1e-9 and scaled_err(ow.H, G["it0_H_full"]) < 1e-9
    assert rel(ow.g, G["it0_g_full"]) < 1e-9
    assert rel(ow.med, G["it0_median_depths_full"]) < 1e-12
    assert (ow.poses - G["it0_kf_poses_new"]).abs().max() < 1e-9
    assert (ow.P - G["it0_P_new"]).abs().max() < 1e-8
    assert rel(delta, G["it0_delta"]) < 1e-6


def _regenerated_window(G, predictor):
    from como_amd import synth
    st = synth.make_window(B=int(G["B"]), H=int(G["H"]), W=int(G["W"]), m=int(G["m"]), dtype=torch.float64, seed=int(G["seed"]),
                           predictor=predictor, aff_noise=float(G["aff_noise"]) if "aff_noise" in G else 0.0,
                           channels=int(G["channels"]) if "channels" in G else 1)
    # same seeds -> same inputs (discrete choices identical; values to the last bits: CPU sin / exp / BLAS depend on the host ISA)
    assert (st["kf_poses"] - (G["kf_poses"] if "kf_poses" in G else G["it0_kf_poses_in"])).abs().max() < 1e-12
    assert (st["coords_m"] - G["coords_m"]).abs().max() < 1e-9 and (st["P_m"] - G["P_m"]).abs().max() < 1e-12
    return st


def _check_iterations(ow, G, iters, tol_pose):
    for it in range(iters):
        delta = ow.iterate()
        g = lambda k: G[f"it{it}_{k}"]
        rid = g("kf_ref_ids")
        assert torch.equal(ow.aux["valid"].sum(dim=1), g("pair_nvalid"))            # per-pair valid counts: exact
        assert abs(float(ow.aux["sigma"]) - float(g("sigma_r"))) <= 2e-9 * float(g("sigma_r"))
        assert rel(ow.med, g("median_depths_full")) < 1e-9
        assert rel(ow.med_subset, g("median_depths_subset")) < 1e-9
        assert rel(torch.diagonal(ow.H), g("H_full_diag")) < 1e-7
        assert rel(ow.g, g("g_full")) < 1e-6
        assert (ow.poses - g("kf_poses_new")).abs().max() < tol_pose
        assert (ow.P - g("P_new")).abs().max() < 100 * tol_pose
        assert len(rid) == ow.aux["valid"].shape[0]


def test_oracle_window32_vs_reference():
    """Config 4: 32 keyframes, 62 pairs (reduced resolution); two reference iterations."""
    G = load_golden("ba_window32_f64.npz")
    st = _regenerated_window(G, lambda cov, cm: depthcov.prep_predictor(cov, cm, 1.0))
    assert rel(st["K_mm_inv"], G["K_mm_inv"]) < 1e-6
    ow = OracleWindow(st, window=int(G["window"]))
    assert ow.D == G["it0_g_full"].shape[0] == 8 * 32 + 3 * st["P_m"].shape[0]
    _check_iterations(ow, G, 2, 1e-8)


def test_oracle_rgb_window_vs_reference():
    """`color: rgb`: 3-channel keyframes through the reference's Mapping.iterate (one median over all (pixel, channel)
    residuals, one affine pair per frame, Gram summed over pixels and channels: photo.py:24-52, 112-128)."""
    G = load_golden("ba_window_rgb_kf_f64.npz")
    st = _regenerated_window(G, lambda cov, cm: depthcov.prep_predictor(cov, cm, 1.0))
    assert st["kf_img_and_grads"].shape[1] == 9
    ow = OracleWindow(st, window=int(G["window"]))
    assert ow.vals.shape[2] == 3
    _check_iterations(ow, G, 2, 1e-8)
    assert scaled_err(ow.H, G["it1_H_full"]) < 1e-7 if "it1_H_full" in G else True


def test_oracle_fullsize_window4_vs_reference():
    """The METRIC configuration (8 keyframes, 640x480, m = 64) at the reference's default sub-selection: the oracle against
    three iterations of the reference's own Mapping.iterate (per-pair valid counts exact, sigma_r, medians, system, update)."""
    G = load_golden("fullsize_window4.npz")
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    st = _regenerated_window(G, lambda cov, cm: depthcov.prep_predictor(cov, cm, 1.0))
    ow = OracleWindow(st, window=4)
    assert ow.D == 760
    ow.iterate()
    g = lambda k: G[f"it0_{k}"]
    assert torch.equal(ow.aux["valid"].sum(dim=1), g("pair_nvalid"))
    assert abs(float(ow.aux["sigma"]) - float(g("sigma_r"))) <= 2e-9 * float(g("sigma_r"))
    pi, ii = G["sample_pair"], G["sample_pix"]
    assert torch.equal(ow.aux["valid"][pi, ii], G["sample_valid"])
    assert (ow.aux["r"][pi, ii] - G["sample_r"]).abs().max() < 1e-9
    assert rel(ow.med, g("median_depths_full")) < 1e-9 and rel(ow.med_subset, g("median_depths_subset")) < 1e-9
    assert scaled_err(ow.H, g("H_full")) < 1e-7
    assert rel(ow.g, g("g_full")) < 1e-7
    assert (ow.poses - g("kf_poses_new")).abs().max() < 1e-8
    assert (ow.P - g("P_new")).abs().max() < 1e-6
