"""Seeded synthetic inputs for tests and bench.py (there are no datasets in this image).

Scene = one textured plane seen by a small screw-motion camera trajectory, so
ground-truth depth and poses are known analytically:
pinhole fx=fy=525*(W/640), c=(W-1)/2,(H-1)/2; KF k translated 0.02*k m along x and
rotated k degrees about y; plane depth 0.9-1.15 (the reference's GP predicts log-depth with a ZERO-mean prior,
depth.py:22-24, so a scene at unit depth is the regime where K~ logz_m interpolates without bias -- at 2.5 m the
row-sum deficit of K~ between inducing points pulls the predicted depth towards 1 and plain GN settles 5 cm
off the ground truth); texture = band-limited sum of 12 sinusoids in plane coordinates, values in [0,1].

Everything here is plain torch (runs on CPU or GPU); nothing here is on the
measured hot path.  The GP predictor (K_mm^-1, K~ = K_nm K_mm^-1) is supplied by
the caller (`predictor=`): the HIP implementation in bench.py / gpu tests, the
oracle in CPU tests.
"""
import math

import torch

from como_amd.geometry.lie_algebra import se3_exp


def intrinsics_for(H, W, dtype=torch.float64, device="cpu"):
    f = 525.0 * (W / 640.0)
    K = torch.tensor([[f, 0.0, (W - 1) / 2.0], [0.0, f, (H - 1) / 2.0], [0.0, 0.0, 1.0]],
                     dtype=dtype, device=device)
    return K


def gt_poses(B, dtype=torch.float64, device="cpu", step=0.02, deg=1.0):
    k = torch.arange(B, dtype=dtype, device=device)
    xi = torch.zeros((B, 6), dtype=dtype, device=device)
    xi[:, 1] = k * (deg * math.pi / 180.0)  # omega_y
    T = se3_exp(xi)
    T[:, 0, 3] = step * k
    T[:, 1, 3] = 0.2 * step * k
    return T


class PlaneScene:
    """Plane n.P = d in world coordinates with a sinusoid texture."""

    def __init__(self, seed=0, dtype=torch.float64, device="cpu", freq_scale=1.0):
        """freq_scale = W/640 keeps the texture's spectrum fixed in PIXEL units at any resolution."""
        g = torch.Generator().manual_seed(seed)
        n = torch.tensor([0.12, -0.07, 1.0], dtype=dtype)
        self.n = (n / n.norm()).to(device)
        self.d = 1.0
        e1 = torch.linalg.cross(self.n.cpu(), torch.tensor([0.0, 1.0, 0.0], dtype=dtype))
        e1 = e1 / e1.norm()
        e2 = torch.linalg.cross(self.n.cpu(), e1)
        self.e1, self.e2 = e1.to(device), e2.to(device)
        nf = 12
        self.freq = (freq_scale * (0.8 + 17.0 * torch.rand((nf, 2), generator=g, dtype=dtype))).to(device)
        self.freq *= torch.where(torch.rand((nf, 2), generator=g) < 0.5, -1.0, 1.0).to(device=device, dtype=dtype)
        self.phase = (2 * math.pi * torch.rand((nf,), generator=g, dtype=dtype)).to(device)
        amp = 0.3 + torch.rand((nf,), generator=g, dtype=dtype)
        self.amp = (amp / amp.sum()).to(device)

    def depth_and_points(self, T_wc, K, H, W):
        """Per-pixel camera depth (H,W) and world points (H,W,3) for one pose."""
        dt, dev = T_wc.dtype, T_wc.device
        v, u = torch.meshgrid(torch.arange(H, dtype=dt, device=dev),
                              torch.arange(W, dtype=dt, device=dev), indexing="ij")
        ray = torch.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)), dim=-1)
        rw = ray @ T_wc[:3, :3].T
        t = T_wc[:3, 3]
        lam = (self.d - (self.n * t).sum()) / (rw * self.n).sum(-1)
        Pw = t + lam[..., None] * rw
        return lam, Pw

    def texture(self, Pw):
        uv = torch.stack(((Pw * self.e1).sum(-1), (Pw * self.e2).sum(-1)), dim=-1)
        arg = 2 * math.pi * (uv[..., None, :] * self.freq).sum(-1) + self.phase
        return 0.5 + 0.5 * (self.amp * torch.sin(arg)).sum(-1)

    def render(self, T_wc, K, H, W):
        z, Pw = self.depth_and_points(T_wc, K, H, W)
        return self.texture(Pw), z


def scharr_and_stack(img):
    """(B,c,H,W) -> (B,3c,H,W) [I | gx | gy] (each c channels): Scharr/32 with reflect padding.

    Same filter as reference como/utils/image_processing.py:8-44 (test/bench
    input construction only; the product kernel is como_amd.utils.image_processing).
    """
    kx = torch.tensor([[-3.0, 0.0, 3.0], [-10.0, 0.0, 10.0], [-3.0, 0.0, 3.0]],
                      dtype=img.dtype, device=img.device) / 32.0
    p = torch.nn.functional.pad(img, (1, 1, 1, 1), mode="reflect")
    c = img.shape[1]
    gx = torch.nn.functional.conv2d(p, kx.view(1, 1, 3, 3).repeat(c, 1, 1, 1), groups=c)
    gy = torch.nn.functional.conv2d(p, kx.T.contiguous().view(1, 1, 3, 3).repeat(c, 1, 1, 1), groups=c)
    return torch.cat((img, gx, gy), dim=1)


def smooth_noise(B, C, H, W, gen, cells=6, dtype=torch.float64):
    low = torch.randn((B, C, cells, cells + 2), generator=gen, dtype=dtype)
    return torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=True)


def synthetic_cov_params(B, H, W, seed=0, dtype=torch.float64, device="cpu"):
    """Synthetic DepthCov network output (B,3,H,W) 'raw' -> 2x2 covariance image (B,4,H,W).

    raw ch0/ch1 ~ -2.5 (+ smooth noise), ch2 ~ smooth noise; then the reference's
    normalisation (gaussian_kernel.py:6-49): x=e^clamp, z=e^clamp, rho=0.99 tanh,
    E = [[x, sqrt(xz-1e-8) rho], [., z]].
    """
    g = torch.Generator().manual_seed(seed + 101)
    raw = smooth_noise(B, 3, H, W, g, dtype=dtype)
    raw[:, 0:2] = -2.5 + 0.3 * raw[:, 0:2]
    raw[:, 2] = 0.3 * raw[:, 2]
    x = torch.exp(torch.clamp(raw[:, 0], math.log(1e-3), math.log(1e4)))
    z = torch.exp(torch.clamp(raw[:, 1], math.log(1e-3), math.log(1e4)))
    off = torch.sqrt(x * z - 1e-8) * (0.99 * torch.tanh(raw[:, 2]))
    E = torch.stack((x, off, off, z), dim=1)
    return E.to(device)


def _thin(points, chosen, min_dist):
    """Greedy: keep points (k,2) that are >= min_dist from everything in `chosen` (list)."""
    keep = []
    for i in range(points.shape[0]):
        p = points[i]
        ok = True
        for q in chosen:
            if (p - q).abs().max() < min_dist:
                ok = False
                break
        if ok:
            chosen.append(p)
            keep.append(i)
    return keep


def build_landmarks(scene, T_gt, K, H, W, m, seed=0, reuse_frac=0.625, border=4):
    """Sequentially add keyframes: reuse visible landmarks, add new grid points up to m per KF.

    Returns coords_m (B,m,2) float (row, col) = pixel of each KF's m landmarks,
    corr (B,L) bool with exactly m per row, P_gt (L,3) world landmarks on the plane.
    Mimics what the reference's correspondence step (frontend/corr.py) hands to
    Mapping: every KF observes exactly m landmarks, some shared with earlier KFs.
    """
    B = T_gt.shape[0]
    g = torch.Generator().manual_seed(seed + 7)
    dt = T_gt.dtype
    Tc = T_gt.cpu()
    Kc = K.cpu().to(dt)
    P_all = []          # list of (3,) tensors
    obs = []            # per KF list of landmark ids
    coords = []
    gy, gx = int(round(math.sqrt(m * H / W))), 0
    gx = int(math.ceil(m / gy))
    cell_h, cell_w = (H - 2 * border) / gy, (W - 2 * border) / gx
    min_dist = 0.45 * min(cell_h, cell_w)
    sc = PlaneScene.__new__(PlaneScene)
    sc.__dict__ = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in scene.__dict__.items()}
    for k in range(B):
        R, t = Tc[k, :3, :3], Tc[k, :3, 3]
        ids, pts, chosen = [], [], []
        if P_all:
            Pw = torch.stack(P_all)
            Pc = (Pw - t) @ R
            u = Kc[0, 0] * Pc[:, 0] / Pc[:, 2] + Kc[0, 2]
            v = Kc[1, 1] * Pc[:, 1] / Pc[:, 2] + Kc[1, 2]
            vis = (Pc[:, 2] > 0.1) & (u >= border) & (u < W - 1 - border) & (v >= border) & (v < H - 1 - border)
            cand = torch.nonzero(vis)[:, 0]
            cand = cand[torch.randperm(cand.numel(), generator=g)]
            rc = torch.stack((v[cand], u[cand]), dim=1)
            keep = _thin(rc, chosen, min_dist)
            keep = keep[: int(reuse_frac * m)]
            chosen = [rc[i] for i in keep]
            ids = [int(cand[i]) for i in keep]
            pts = [rc[i] for i in keep]
        # new points on a jittered grid
        jit = torch.rand((gy, gx, 2), generator=g, dtype=dt) * 0.6 + 0.2
        cells = torch.stack(torch.meshgrid(torch.arange(gy, dtype=dt), torch.arange(gx, dtype=dt), indexing="ij"), -1)
        grid = (cells + jit) * torch.tensor([cell_h, cell_w], dtype=dt) + border
        grid = torch.floor(grid.reshape(-1, 2))  # integer pixel centres
        grid = grid[torch.randperm(grid.shape[0], generator=g)]
        md = min_dist
        while len(pts) < m:
            keep = _thin(grid, chosen, md)
            for i in keep:
                if len(pts) >= m:
                    break
                rcp = grid[i]
                ray = torch.stack(((rcp[1] - Kc[0, 2]) / Kc[0, 0], (rcp[0] - Kc[1, 2]) / Kc[1, 1], torch.ones((), dtype=dt)))
                rw = R @ ray
                lam = (sc.d - (sc.n * t).sum()) / (rw * sc.n).sum()
                P_all.append(t + lam * rw)
                ids.append(len(P_all) - 1)
                pts.append(rcp)
            md *= 0.7
            if md < 1.0 and len(pts) < m:
                raise RuntimeError("could not place inducing points")
        obs.append(ids)
        coords.append(torch.stack(pts))
    L = len(P_all)
    corr = torch.zeros((B, L), dtype=torch.bool)
    coords_sorted = []
    for k in range(B):
        order = sorted(range(m), key=lambda i: obs[k][i])  # batched order = ascending landmark id
        corr[k, [obs[k][i] for i in order]] = True
        coords_sorted.append(coords[k][order])
    return torch.stack(coords_sorted), corr, torch.stack(P_all)


def make_recent(ts, H, W, seed, dtype=torch.float64, device="cpu", channels=1, pose_noise=1e-3, aff_noise=0.02):
    """One-way frames at (fractional) keyframe times `ts` of the make_window(seed) scene: images (same textures, channel for
    channel), perturbed poses and affine parameters -- what the reference Mapping holds in its recent_* members."""
    g = torch.Generator().manual_seed(seed + 5)
    K64 = intrinsics_for(H, W, torch.float64)
    scenes = [PlaneScene(seed=seed + 1000 * ch, dtype=torch.float64, device="cpu", freq_scale=W / 640.0) for ch in range(channels)]
    imgs, poses = [], []
    for t in ts:
        xi = torch.zeros(1, 6, dtype=torch.float64)
        xi[0, 1] = t * math.pi / 180.0
        Tr = se3_exp(xi)[0]
        Tr[0, 3] = 0.02 * t
        Tr[1, 3] = 0.004 * t
        imgs.append(torch.stack([sc.render(Tr, K64, H, W)[0] for sc in scenes]))
        poses.append(Tr @ se3_exp(pose_noise * torch.randn((1, 6), generator=g, dtype=torch.float64))[0])
    aff = aff_noise * torch.randn((len(ts), 2, 1), generator=g, dtype=torch.float64)
    return {"recent_img_and_grads": scharr_and_stack(torch.stack(imgs).to(dtype)).to(device),
            "recent_poses": torch.stack(poses).to(dtype).to(device), "recent_aff_params": aff.to(dtype).to(device),
            "recent_timestamps": torch.tensor(list(ts), dtype=dtype, device=device)}


def make_window(B=8, H=480, W=640, m=64, dtype=torch.float64, device="cpu", seed=0,
                predictor=None, pose_noise=1e-3, depth_noise=0.02, aff_noise=0.0, channels=1):
    """Build one keyframe-window state (what reference Mapping holds before iterate()).
    channels = 3: colour images (`color: rgb`): channel 0 is the gray scene's texture, channels 1 and 2 are two more
    textures on the same plane.

    predictor(cov_params_img (B,4,H,W), coords_m (B,m,2)) -> (K_mm_inv, L_mm, Knm_Kmminv (B,H,W,m))
    """
    g = torch.Generator().manual_seed(seed)
    scene = PlaneScene(seed=seed, dtype=torch.float64, device="cpu", freq_scale=W / 640.0)
    K64 = intrinsics_for(H, W, torch.float64)
    T_gt = gt_poses(B, torch.float64)
    imgs, depths = [], []
    for k in range(B):
        I, z = scene.render(T_gt[k], K64, H, W)
        imgs.append(I)
        depths.append(z)
    img = torch.stack(imgs)[:, None]
    for ch in range(1, channels):
        tex = PlaneScene(seed=seed + 1000 * ch, dtype=torch.float64, device="cpu", freq_scale=W / 640.0)
        img = torch.cat((img, torch.stack([tex.render(T_gt[k], K64, H, W)[0] for k in range(B)])[:, None]), dim=1)
    img = img + 0.002 * torch.randn(img.shape, generator=g, dtype=torch.float64)
    depth_gt = torch.stack(depths)
    img_and_grads = scharr_and_stack(img.to(dtype))
    cov = synthetic_cov_params(B, H, W, seed=seed, dtype=torch.float64).to(dtype)
    coords_m, corr, P_gt = build_landmarks(scene, T_gt, K64, H, W, m, seed=seed)
    # perturbed initial estimates; KF0 stays at GT (it is the gauge anchor)
    xi = pose_noise * torch.randn((B, 6), generator=g, dtype=torch.float64)
    xi[0] = 0
    T0 = T_gt @ se3_exp(xi)
    # perturb landmarks along the viewing ray of their first observer (log-depth noise)
    first = torch.argmax(corr.int(), dim=0)
    ids_b = torch.stack([torch.nonzero(corr[k])[:, 0] for k in range(B)])
    t_first = T_gt[first, :3, 3]
    s = torch.exp(depth_noise * torch.randn((P_gt.shape[0], 1), generator=g, dtype=torch.float64))
    P0 = t_first + s * (P_gt - t_first)
    aff = aff_noise * torch.randn((B, 2, 1), generator=g, dtype=torch.float64)
    aff[0] = 0
    st = {
        "intrinsics": K64.to(dtype)[None].to(device),
        "kf_poses": T0.to(dtype).to(device),
        "kf_aff_params": aff.to(dtype).to(device),
        "kf_img_and_grads": img_and_grads.to(device),
        "cov_params_img": cov.to(device),
        "coords_m": coords_m.to(dtype).to(device),
        "correspondence_mask": corr.to(device),
        "P_m": P0.to(dtype).to(device),
        "poses_gt": T_gt.to(dtype).to(device),
        "P_gt": P_gt.to(dtype).to(device),
        "depth_gt": depth_gt.to(dtype).to(device),
        "kf_timestamps": torch.arange(B, dtype=dtype, device=device),
        # landmark first seen in this KF (reference Mapping.py:306-312) and its first-observation pixel (x, y)
        "obs_ref_mask": (first[ids_b] == torch.arange(B)[:, None]).to(device),
        "pm_first_obs": coords_m.flip(-1).to(dtype).to(device),
        # gauge anchors (reference Mapping.py:275-280, 355-367): first pose and the landmarks of the oldest keyframe
        "pose_anchor": T_gt[0:1].to(dtype).to(device),
        "P_anchor": P_gt[corr[0]].to(dtype).to(device),
        "median_depth_init": torch.full((B,), float(scene.d), dtype=dtype, device=device),
    }
    if predictor is not None:
        K_mm_inv, L_mm, Kt = predictor(st["cov_params_img"], st["coords_m"])
        st.update({"K_mm_inv": K_mm_inv, "L_mm": L_mm, "Knm_Kmminv": Kt})
    return st


def make_tracking_pair(H=480, W=640, dtype=torch.float32, device="cpu", seed=0, levels=3,
                       pose_noise=1e-3, channels=1):
    """Reference keyframe (image, GT depth) + a second frame, for 2-frame tracking (config 2).
    channels = 3: colour images (the gray texture + two more textures on the same plane)."""
    g = torch.Generator().manual_seed(seed)
    scene = PlaneScene(seed=seed, freq_scale=W / 640.0)
    K = intrinsics_for(H, W)
    T = gt_poses(2, step=0.012, deg=0.7)
    I0, z0 = scene.render(T[0], K, H, W)
    I1, _ = scene.render(T[1], K, H, W)
    if channels > 1:
        tex = [PlaneScene(seed=seed + 1000 * ch, freq_scale=W / 640.0) for ch in range(1, channels)]
        I0 = torch.stack([I0] + [t.render(T[0], K, H, W)[0] for t in tex])
        I1 = torch.stack([I1] + [t.render(T[1], K, H, W)[0] for t in tex])
    I0 = I0 + 0.002 * torch.randn(I0.shape, generator=g, dtype=torch.float64)
    I1 = 0.97 * (I1 + 0.002 * torch.randn(I1.shape, generator=g, dtype=torch.float64)) + 0.01
    T10_gt = torch.linalg.inv(T[1]) @ T[0]  # T_ji: i = ref (0), j = current (1)
    xi = pose_noise * torch.randn((1, 6), generator=g, dtype=torch.float64)
    T10_init = T10_gt[None] @ se3_exp(xi)
    return {
        "img_ref": (I0[None, None] if channels == 1 else I0[None]).to(dtype).to(device),
        "depth_ref": z0[None, None].to(dtype).to(device),
        "img_cur": (I1[None, None] if channels == 1 else I1[None]).to(dtype).to(device),
        "intrinsics": K.to(dtype).to(device),
        "Tji_gt": T10_gt[None].to(dtype).to(device),
        "Tji_init": T10_init.to(dtype).to(device),
        "levels": levels,
    }


def depthcov_state_dict(seed=0, num_levels=5, base=16, feature_channels=3, device="cpu"):
    """Seeded float32 parameters for the DepthCov network under the reference's state_dict key names
    (como/depth_cov/core/DepthCovModule.py:22-31 -> UNet(num_levels=5, in=3, base=16, feature=3, k=3)).  There is no
    checkpoint in this environment: weights are He-scaled normal draws from numpy's RandomState (bit-reproducible),
    GroupNorm gains near 1."""
    import numpy as np
    rs = np.random.RandomState(seed)
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = torch.from_numpy((rs.randn(cout, cin, k, k) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy((0.05 * rs.randn(cout)).astype(np.float32))

    def resblock(prefix, cin, cout):
        conv(prefix + "conv1", cout, cin, 3)
        conv(prefix + "conv2", cout, cout, 3)
        conv(prefix + "conv3", cout, cin, 1)
        sd[prefix + "norm.weight"] = torch.from_numpy((1.0 + 0.1 * rs.randn(cout)).astype(np.float32))
        sd[prefix + "norm.bias"] = torch.from_numpy((0.1 * rs.randn(cout)).astype(np.float32))

    p = "gaussian_cov_net."
    resblock(p + "base.", 3, base)
    c = base
    for i in range(num_levels):
        resblock(f"{p}down_convs.{i}.conv_block.", c, 2 * c)
        conv(f"{p}up_convs.{i}.upsample.1", c, 2 * c, 3)
        resblock(f"{p}up_convs.{i}.conv_block.", 2 * c, c)
        if i < num_levels - 1:
            conv(f"{p}feature_convs.{i}", feature_channels, c, 1)
        c *= 2
    for i in range(num_levels - 1):
        sd[f"cov_modules.{i}.scale_param"] = torch.tensor(0.0)
        sd[f"log_depth_var_scales.{i}"] = torch.tensor(0.0)
    return {k: v.to(device) for k, v in sd.items()}
