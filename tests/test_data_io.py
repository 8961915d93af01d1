"""CPU: the sequence readers and ground-truth converters of como_amd/data on synthetic directory trees
(no dataset exists in this image): index parsing, timestamps, intrinsics (against values computed with the reference's own
`resize_intrinsics`, tests/golden/dataset_intrinsics.npz), decode + resize conventions, TUM trajectory round trip, ATE helpers."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN


def _write_jpg(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path, quality=95)


def _png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


@pytest.fixture
def rng():
    return np.random.default_rng(0)


def test_replica_reader_and_gt_converter(tmp_path, rng):
    from como_amd.data.gt_convert import convert_replica_traj
    from como_amd.data.odom_datasets import ReplicaDataset, get_dataset
    from como_amd.utils.io import tq_to_pose
    root = tmp_path / "replica" / "room0"
    (root / "results").mkdir(parents=True)
    imgs = [rng.integers(0, 255, (680, 1200, 3), dtype=np.uint8) for _ in range(3)]
    for i, im in enumerate(imgs):
        _write_jpg(str(root / "results" / f"frame{i:06d}.jpg"), im)
    seq = str(root) + "/"
    ds = get_dataset("replica", [192, 256], seq)
    assert isinstance(ds, ReplicaDataset) and len(ds) == 3 and ds.save_traj_name.endswith("room0")
    ts, rgb = ds[2]
    assert ts == pytest.approx(2 / 30.0) and rgb.shape == (3, 192, 256) and rgb.dtype == torch.float32
    assert 0.0 <= float(rgb.min()) and float(rgb.max()) <= 1.0
    G = np.load(os.path.join(GOLDEN, "dataset_intrinsics.npz"))
    assert np.allclose(ds.intrinsics.numpy(), G["replica"], rtol=0, atol=1e-5)
    # ground truth: flattened 4x4 per line -> TUM file that reads back to the same poses
    T = np.tile(np.eye(4), (5, 1, 1))
    T[:, :3, 3] = rng.normal(size=(5, 3))
    th = 0.3
    T[:, :3, :3] = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    np.savetxt(seq + "traj.txt", T.reshape(5, 16))
    out = convert_replica_traj(seq)
    rows = np.loadtxt(out)
    assert rows.shape == (5, 8) and np.allclose(rows[:, 0], np.arange(5) / 30.0, atol=1e-4)
    assert np.abs(tq_to_pose(rows[:, 1:]) - T).max() < 2e-4                      # four decimals in the file


def test_scannet_reader_and_gt_converter(tmp_path, rng):
    from como_amd.data.gt_convert import convert_scannet_traj
    from como_amd.data.odom_datasets import ScanNetOdometryDataset
    root = tmp_path / "scannet" / "scans_test" / "scene0707_00"
    (root / "color").mkdir(parents=True)
    (root / "pose").mkdir()
    for i in (0, 1, 2, 10):
        _write_jpg(str(root / "color" / f"{i}.jpg"), rng.integers(0, 255, (480, 640, 3), dtype=np.uint8))
    info = ["appVersionId = 1", "colorHeight = 968", "colorToDepthExtrinsics = 1 0 0 0", "colorWidth = 1296", "depthHeight = 480",
            "depthWidth = 640", "fx_color = 1170.187988", "fx_depth = 570", "fy_color = 1170.187988", "fy_depth = 570",
            "mx_color = 647.75", "mx_depth = 319.5", "my_color = 483.75", "my_depth = 239.5"]
    (root / "scene0707_00.txt").write_text("\n".join(info) + "\n")
    seq = str(root) + "/"
    ds = ScanNetOdometryDataset(seq, [192, 256], crop_size=10)
    assert len(ds) == 4 and [os.path.basename(p) for p in ds.rgb_list] == ["0.jpg", "1.jpg", "2.jpg", "10.jpg"]     # numeric order
    ts, rgb = ds[3]
    assert ts == pytest.approx(0.1) and rgb.shape == (3, 192, 256)
    G = np.load(os.path.join(GOLDEN, "dataset_intrinsics.npz"))
    assert np.allclose(ds.intrinsics.numpy(), G["scannet"], rtol=0, atol=1e-4)
    for i in range(4):
        P = np.eye(4)
        P[0, 3] = i
        if i == 2:
            P[0, 0] = -np.inf                                                    # ScanNet marks lost frames with -inf
        np.savetxt(str(root / "pose" / f"{i}.txt"), P)
    rows = np.loadtxt(convert_scannet_traj(seq))
    assert rows.shape == (3, 8) and np.allclose(rows[:, 0], [0.0, 1 / 30.0, 3 / 30.0], atol=1e-4) and np.allclose(rows[:, 1], [0, 1, 3])


def test_tum_reader_index_intrinsics_and_rectification(tmp_path, rng):
    from como_amd.data import odom_datasets as od
    root = tmp_path / "tum" / "rgbd_dataset_freiburg3_long_office_household"
    (root / "rgb").mkdir(parents=True)
    names = ["1341847980.722988.png", "1341847980.754743.png"]
    for nm in names:
        _png(str(root / "rgb" / nm), rng.integers(0, 255, (480, 640, 3), dtype=np.uint8))
    (root / "rgb.txt").write_text("# color images\n# file: x.bag\n# timestamp filename\n" +
                                  "".join(f"{nm[:-4]} rgb/{nm}\n" for nm in names))
    seq = str(root) + "/"
    ds = od.get_dataset("tum", [192, 256], seq)
    assert len(ds) == 2 and ds.load_timestamp(1) == pytest.approx(1341847980.754743)
    G = np.load(os.path.join(GOLDEN, "dataset_intrinsics.npz"))
    assert np.allclose(ds.intrinsics.numpy(), G["tum3"], atol=1e-4)
    ts, rgb = ds[0]
    assert rgb.shape == (3, 192, 256)
    # rectification helpers: zero distortion -> the identity map and the original camera matrix (up to the (w-1) viewport fit)
    K = np.array([[520.9, 0, 325.1], [0, 521.0, 249.7], [0, 0, 1.0]])
    m1, m2 = od.undistort_rectify_map(K, [0, 0, 0, 0, 0], K, (640, 480))
    u, v = np.meshgrid(np.arange(640), np.arange(480))
    assert np.abs(m1 - u).max() < 1e-3 and np.abs(m2 - v).max() < 1e-3
    # distort / undistort are inverse on the normalised plane for the freiburg2 coefficients
    dist = od.TumOdometryDataset.CAMERAS[2][1]
    xs, ys = np.meshgrid(np.linspace(-0.55, 0.55, 9), np.linspace(-0.4, 0.4, 7))
    xd, yd = od._distort_normalised(xs, ys, dist)
    xu, yu = od._undistort_normalised(xd, yd, dist, iters=20)
    assert np.abs(xu - xs).max() < 1e-6 and np.abs(yu - ys).max() < 1e-6
    Ku = od.optimal_new_camera_matrix_alpha0(K, dist, (640, 480))
    m1, m2 = od.undistort_rectify_map(K, dist, Ku, (640, 480))
    assert m1.min() >= -1.0 and m1.max() <= 640.0 and m2.min() >= -1.0 and m2.max() <= 480.0        # alpha = 0: no invalid border
    img = torch.rand(3, 480, 640)
    out = od.remap_bilinear(img, m1, m2)
    assert out.shape == img.shape and torch.isfinite(out).all()
    ds1 = od.TumOdometryDataset(seq.replace("freiburg3", "freiburg3"), [480, 640])
    assert ds1.map1 is None


def test_ate_helpers():
    from como_amd.utils.ate import ate_rmse, associate, umeyama
    rng = np.random.default_rng(1)
    P = np.tile(np.eye(4), (20, 1, 1))
    P[:, :3, 3] = np.cumsum(rng.normal(size=(20, 3)) * 0.1, axis=0)
    assert ate_rmse(P, P) == 0.0
    th = 0.4
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    Q = P.copy()
    Q[:, :3, 3] = (2.5 * (R @ P[:, :3, 3].T)).T + np.array([0.3, -1.0, 2.0])
    assert ate_rmse(Q, P) > 0.1 and ate_rmse(Q, P, "sim3") < 1e-12
    s, Rr, t = umeyama(P[:, :3, 3], Q[:, :3, 3], True)
    assert abs(s - 2.5) < 1e-12 and np.abs(Rr - R).max() < 1e-12
    assert ate_rmse(Q, P, "se3") > 1e-3                                          # a scale change is not an SE(3) alignment
    assert associate([0.0, 0.1, 0.2], [0.101, 0.3, 0.001]) == [(0, 2), (1, 0)]


def test_runner_config_and_keyframe_history(tmp_path):
    """como_amd.run: the YAML of config/como.yml parses into the two sections `ComoSeq` takes (one device, the reference's
    constants); `device: cpu` is refused; `KeyframeHistory` is GuiWindow.update_kf_vars (:329-357) for timestamps / poses."""
    import yaml
    from como_amd import run
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = run.load_slam_cfg(os.path.join(root, "config", "como.yml"), device="cuda:3")
    assert cfg["tracking"]["device"] == cfg["mapping"]["device"] == "cuda:3"
    assert cfg["mapping"]["graph"] == {"num_keyframes": 9, "num_one_way_frames": 24}
    assert cfg["mapping"]["sampling"]["max_num_coords"] == 64 and cfg["mapping"]["photo_construction"]["nonmax_suppression_window"] == 4
    assert cfg["tracking"]["term_criteria"] == {"max_iter": 50, "delta_norm": 1e-3, "rel_tol": 1e-3, "grad_norm": 1.0}
    assert cfg["mapping"]["pix_dtype"] == "double" and cfg["mapping"]["network_size"] == [192, 256] and cfg["mapping"]["viewer_snapshots"] is False
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        run.load_slam_cfg(os.path.join(root, "config", "como.yml"), device="cpu")
    bad = tmp_path / "bad.yml"
    bad.write_text(yaml.safe_dump({"tracking": {}}))
    with pytest.raises(ValueError, match="mapping"):
        run.load_slam_cfg(str(bad))
    with pytest.raises(FileNotFoundError, match="random_weights"):
        run.load_model({"model_path": str(tmp_path / "none.ckpt")}, "cpu")
    # the history: window of 3; keyframes 1..5 inserted one at a time, poses refined while in the window
    h = run.KeyframeHistory()
    P = lambda v, n: torch.full((n, 4, 4), float(v))
    h.update([1.0], P(10, 1))
    h.update([1.0, 2.0], P(11, 2))
    h.update([1.0, 2.0, 3.0], P(12, 3))
    h.update([2.0, 3.0, 4.0], P(13, 3))            # keyframe 1 left the window: its last pose (12) stays
    h.update([2.0, 3.0, 4.0], P(14, 3))
    h.update([3.0, 4.0, 5.0], P(15, 3))
    assert h.timestamps == [1.0, 2.0, 3.0, 4.0, 5.0]
    assert [float(p[0, 0]) for p in h.poses] == [12.0, 14.0, 15.0, 15.0, 15.0]
