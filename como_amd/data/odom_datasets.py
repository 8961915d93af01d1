"""RGB sequence readers with the reference's interface (como/data/odom_datasets.py:21-307, dataset_factory.py:11-25):
`dataset[i] -> (timestamp, rgb (3,H,W) float32 in [0,1])`, `dataset.intrinsics` (3,3) for the resized images,
`dataset.save_traj_name`.  Image decoding through PIL (the reference uses OpenCV, which this image does not have); the
bilinear resize is `cv2.resize(INTER_LINEAR)`'s half-pixel-centre convention = `F.interpolate(align_corners=False)` without
antialiasing.  OpenCV arithmetic is third-party to the reference, so these readers are **parity-unpinned** wherever OpenCV
does more than decode: the TUM freiburg1 / freiburg2 rectification (`getOptimalNewCameraMatrix(alpha=0)` +
`initUndistortRectifyMap` + `remap`) is restated from OpenCV's documented algorithm below."""
import glob
import os
import re

import numpy as np
import torch
import torch.nn.functional as F

from como_amd.geometry.camera import resize_intrinsics


def _imread_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def _to_tensor(rgb_u8):
    return torch.from_numpy(np.array(rgb_u8, copy=True)).permute(2, 0, 1).float() / 255.0                  # TF.to_tensor


def _resize_bilinear(rgb, size):
    return F.interpolate(rgb[None], size=list(size), mode="bilinear", align_corners=False, antialias=False)[0]


def odom_collate_fn(batch):
    assert len(batch) == 1
    return batch[0][0], batch[0][1].unsqueeze(0)


class OdometryDataset:
    def __init__(self, img_size):
        self.is_live = False
        self.img_size = list(img_size)

    def __len__(self):
        return self.data_len

    def __getitem__(self, idx):
        return self.load_timestamp(idx), self.load_rgb(idx)


# ---- OpenCV's rectification, restated (modules/calib3d: undistortPoints, getOptimalNewCameraMatrix, initUndistortRectifyMap) ----
def _distort_normalised(x, y, dist):
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return xd, yd


def _undistort_normalised(xd, yd, dist, iters=5):
    """cvUndistortPoints' fixed-point iteration (5 iterations, the default criteria)."""
    k1, k2, p1, p2, k3 = dist
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + r2 * (k1 + r2 * (k2 + r2 * k3)))
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (xd - dx) * icdist
        y = (yd - dy) * icdist
    return x, y


def optimal_new_camera_matrix_alpha0(K, dist, size_wh):
    """getOptimalNewCameraMatrix(K, dist, size, alpha=0, newImgSize=size): the largest axis-aligned rectangle of the normalised
    plane whose 9 x 9 undistorted border grid lies inside the image, mapped onto the viewport."""
    w, h = size_wh
    N = 9
    gx, gy = np.meshgrid(np.arange(N) * (w - 1) / (N - 1.0), np.arange(N) * (h - 1) / (N - 1.0))
    xn, yn = _undistort_normalised((gx - K[0, 2]) / K[0, 0], (gy - K[1, 2]) / K[1, 1], dist)
    ix0, ix1 = xn[:, 0].max(), xn[:, -1].min()
    iy0, iy1 = yn[0, :].max(), yn[-1, :].min()
    fx, fy = (w - 1) / (ix1 - ix0), (h - 1) / (iy1 - iy0)
    return np.array([[fx, 0.0, -fx * ix0], [0.0, fy, -fy * iy0], [0.0, 0.0, 1.0]])


def undistort_rectify_map(K, dist, K_new, size_wh):
    """initUndistortRectifyMap(K, dist, None, K_new, size, CV_32FC1): source pixel (map1, map2) of every destination pixel."""
    w, h = size_wh
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    xd, yd = _distort_normalised((u - K_new[0, 2]) / K_new[0, 0], (v - K_new[1, 2]) / K_new[1, 1], dist)
    return (xd * K[0, 0] + K[0, 2]).astype(np.float32), (yd * K[1, 1] + K[1, 2]).astype(np.float32)


def remap_bilinear(rgb, map1, map2):
    """cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) of a (3,H,W) float tensor."""
    H, W = rgb.shape[-2:]
    gx = (torch.from_numpy(map1) + 0.5) / W * 2 - 1
    gy = (torch.from_numpy(map2) + 0.5) / H * 2 - 1
    grid = torch.stack((gx, gy), dim=-1)[None].to(rgb.dtype)
    return F.grid_sample(rgb[None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0]


class TumOdometryDataset(OdometryDataset):
    """odom_datasets.py:35-159: rgb.txt (three header lines), camera parameters by `freiburg<k>` in the path."""
    CAMERAS = {1: ([[517.3, 0.0, 318.6], [0.0, 516.5, 255.3], [0.0, 0.0, 1.0]], [0.2624, -0.9531, -0.0054, 0.0026, 1.1633]),
               2: ([[520.9, 0.0, 325.1], [0.0, 521.0, 249.7], [0.0, 0.0, 1.0]], [0.2312, -0.7849, -0.0033, -0.0001, 0.9172]),
               3: ([[535.4, 0.0, 320.1], [0.0, 539.2, 247.6], [0.0, 0.0, 1.0]], None)}

    def __init__(self, seq_path, img_size):
        super().__init__(img_size)
        self.seq_path = seq_path
        tmp = seq_path.rsplit("/", 3)
        self.save_traj_name = tmp[1] + "_" + tmp[2]
        with open(seq_path + "rgb.txt") as f:
            lines = f.readlines()
        self.ts_list, self.rgb_list = [], []
        for line in lines[3:]:
            parts = line.split()
            self.ts_list.append(float(parts[0]))
            self.rgb_list.append(os.path.join(seq_path, parts[1]))
        self.data_len = len(self.rgb_list)
        m = re.search(r"freiburg(\d+)", seq_path)
        if m is None or int(m.group(1)) not in self.CAMERAS:
            raise ValueError("TumOdometryDataset: cannot tell the camera (freiburg1/2/3) from " + seq_path)
        self.setup_camera_vars(int(m.group(1)))

    def setup_camera_vars(self, dataset_ind):
        size_orig = torch.tensor([480, 640])
        scale = torch.tensor(self.img_size) / size_orig
        K, dist = self.CAMERAS[dataset_ind]
        K = np.asarray(K, dtype=np.float64)
        if dist is not None:
            K_u = optimal_new_camera_matrix_alpha0(K, dist, (640, 480))
            self.map1, self.map2 = undistort_rectify_map(K, dist, K_u, (640, 480))
            K = K_u
        else:
            self.map1 = self.map2 = None
        self.intrinsics = resize_intrinsics(torch.from_numpy(K).float() if dist is None else torch.from_numpy(K), scale)

    def load_rgb(self, idx):
        rgb = _to_tensor(_imread_rgb(self.rgb_list[idx]))
        if self.map1 is not None:
            rgb = remap_bilinear(rgb, self.map1, self.map2)
        return _resize_bilinear(rgb, self.img_size)

    def load_timestamp(self, idx):
        return self.ts_list[idx]


class ScanNetOdometryDataset(OdometryDataset):
    """odom_datasets.py:162-260: color/<i>.jpg in numeric order, intrinsics from <scene>.txt, border crop, antialiased resize."""

    def __init__(self, seq_path, img_size, crop_size=0):
        super().__init__(img_size)
        self.seq_path, self.crop_size = seq_path, crop_size
        tmp = seq_path.rsplit("/", 4)
        scene_id = tmp[-2]
        self.save_traj_name = tmp[1] + "_" + scene_id
        files = [os.path.join(seq_path + "color/", f) for f in os.listdir(seq_path + "color/") if f.endswith(".jpg")]
        self.rgb_list = sorted(files, key=lambda x: int(re.findall(r"\d+", x.rsplit("/", 1)[-1])[0]))
        with open(seq_path + scene_id + ".txt") as f:
            lines = f.readlines()
        off = 0 if re.match(r"appVersionId", lines[0]) else -1
        val = lambda k: float(np.array(lines[k + off].split(" = ")[1].split(), dtype=np.float64)[0])
        size_orig = torch.tensor([val(1), val(3)])                     # colorHeight, colorWidth
        K = torch.tensor([[val(6), 0.0, val(10)], [0.0, val(8), val(12)], [0.0, 0.0, 1.0]])
        self.intrinsics = resize_intrinsics(K, torch.tensor([480, 640]) / size_orig)        # images are stored at 480 x 640
        self.intrinsics[0, 2] -= crop_size
        self.intrinsics[1, 2] -= crop_size
        self.intrinsics = resize_intrinsics(self.intrinsics, torch.tensor(self.img_size) / torch.tensor([480 - 2 * crop_size,
                                                                                                        640 - 2 * crop_size]))
        self.data_len = len(self.rgb_list)

    def load_rgb(self, idx):
        rgb = _to_tensor(_imread_rgb(self.rgb_list[idx]))
        c = self.crop_size
        h, w = rgb.shape[-2:]
        rgb = rgb[..., c:h - c, c:w - c]
        return F.interpolate(rgb[None], size=self.img_size, mode="bilinear", align_corners=False, antialias=True)[0]

    def load_timestamp(self, idx):
        return idx / 30.0


class ReplicaDataset(OdometryDataset):
    """odom_datasets.py:262-307: results/*.jpg, 1200 x 680 pinhole camera (f = 600)."""

    def __init__(self, seq_path, img_size):
        super().__init__(img_size)
        self.seq_path = seq_path
        tmp = seq_path.rsplit("/", 4)
        self.save_traj_name = tmp[1] + "_" + tmp[-2]
        self.rgb_list = sorted(glob.glob(os.path.join(seq_path, "results/*.jpg")))
        self.data_len = len(self.rgb_list)
        K = torch.tensor([[600.0, 0.0, 599.5], [0.0, 600.0, 339.5], [0.0, 0.0, 1.0]])
        self.intrinsics = resize_intrinsics(K, torch.tensor(self.img_size) / torch.tensor([680, 1200]))

    def load_rgb(self, idx):
        return _resize_bilinear(_to_tensor(_imread_rgb(self.rgb_list[idx])), self.img_size)

    def load_timestamp(self, idx):
        return idx / 30.0


def get_dataset(dataset_type, img_size, dataset_dir):
    """dataset_factory.py:11-25 (no RealSense: live capture is out of scope)."""
    if dataset_type == "replica":
        return ReplicaDataset(dataset_dir, img_size)
    if dataset_type == "tum":
        return TumOdometryDataset(dataset_dir, img_size)
    if dataset_type == "scannet":
        return ScanNetOdometryDataset(dataset_dir, img_size)
    raise ValueError("dataset_type mode: " + dataset_type + " is not implemented.")
