"""Quick kernel timing of the BA chain on random (shape-correct) inputs: 8 KFs, 14 pairs, 640x480, m = 64.
Not the bench.py contract -- an early look at kernel times (torch.cuda events on the current stream)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import como_amd.odom.backend.photo as photo  # noqa: E402

dev = "cuda:0"


def run(dtype, window, iters=10):
    torch.manual_seed(0)
    B, Hh, Ww, m = 8, 480, 640, 64
    n = (Hh // window) * (Ww // window)
    K = torch.tensor([[525., 0, 319.5], [0, 525., 239.5], [0, 0, 1]], dtype=dtype, device=dev)
    v, u = torch.meshgrid(torch.arange(0, Hh, window, dtype=dtype, device=dev), torch.arange(0, Ww, window, dtype=dtype, device=dev), indexing="ij")
    z = 2.5 + 0.2 * torch.rand((B, n), dtype=dtype, device=dev)
    ray = torch.stack(((u.reshape(-1) - 319.5) / 525, (v.reshape(-1) - 239.5) / 525, torch.ones(n, dtype=dtype, device=dev)), -1)
    poses = torch.eye(4, dtype=dtype, device=dev).repeat(B, 1, 1)
    poses[:, 0, 3] = 0.05 * torch.arange(B, dtype=dtype, device=dev)
    Pwn = ((z[..., None] * ray[None]) + poses[:, None, :3, 3]).permute(0, 2, 1).contiguous()
    vals = torch.rand((B, n), dtype=dtype, device=dev)
    dT = torch.randn((B, 18, n), dtype=dtype, device=dev)
    uvec = torch.randn((B, 3, n), dtype=dtype, device=dev)
    Kt = torch.rand((B, Hh * Ww, m), dtype=dtype, device=dev) / m
    pix = ((v.reshape(-1) * Ww + u.reshape(-1)).to(torch.int32))[None].repeat(B, 1).contiguous()
    invz = 0.4 * torch.ones((B, m), dtype=dtype, device=dev)
    dzdP = torch.randn((B, 3), dtype=dtype, device=dev)
    img = torch.rand((B, 3, Hh, Ww), dtype=dtype, device=dev)
    aff = torch.zeros((B, 2), dtype=dtype, device=dev)
    L = 232
    D = 8 * B + 3 * L
    kf_inds = torch.arange(8 * B, device=dev).reshape(B, 8)
    lm = torch.stack([torch.randperm(L, device=dev)[:m].sort().values for _ in range(B)])
    landmark_inds = (3 * lm.repeat_interleave(3, dim=1) + torch.arange(3, device=dev).repeat(m)[None]) + 8 * B
    ref = list(range(0, B - 1)) + list(range(1, B))
    tgt = list(range(1, B)) + list(range(0, B - 1))
    table = photo.PairTable(ref, tgt, [False] * len(ref), B, kf_inds, torch.empty(0, device=dev), landmark_inds, 3 * Hh * Ww, 0, dev)
    H = torch.zeros((D, D), dtype=torch.float64, device=dev)
    g = torch.zeros(D, dtype=torch.float64, device=dev)
    e = torch.zeros((), dtype=torch.float64, device=dev)
    res = {}
    for name, phase in (("all", 0xFF), ("residual+select", 0x3F), ("blocks", 64), ("assemble", 128)):
        def call():
            photo.photo_system_factored(table, poses_all=poses, aff_all=aff, Pwn=Pwn, vals=vals, dPwn_dTwc=dT, uvec=uvec, Kt=Kt,
                                        pixidx=pix, invz=invz, dzdP=dzdP, img_base=img, K=K, H_img=Hh, W_img=Ww, H=H, g=g,
                                        err_out=e, phase=phase)
        call(); call()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            call()
        t1.record()
        torch.cuda.synchronize()
        res[name] = t0.elapsed_time(t1) / iters * 1e3
    pp = len(ref) * n
    res["pixel_pairs"] = pp
    res["blocks_GBps_algorithmic"] = pp * 98 * (4 if dtype == torch.float32 else 8) / (res["blocks"] * 1e-6) / 1e9
    res["chunks"] = photo.last_aux["chunks"]
    res["sigma"] = photo.last_aux["sigma"].tolist()
    return res


if __name__ == "__main__":
    out = {}
    sel = [a for a in sys.argv[1:] if not (a.startswith("v=") or a.startswith("s="))]
    for a in sys.argv[1:]:
        if a.startswith("v="):
            photo.BLOCK_VARIANT = int(a[2:])
        if a.startswith("s="):
            photo.BLOCK_STAGGER = int(a[2:])
    for dtype, nm in ((torch.float32, "f32"), (torch.float64, "f64")):
        for window in (1, 4):
            if sel and f"{nm}_w{window}" not in sel:
                continue
            out[f"{nm}_w{window}"] = run(dtype, window)
            print(nm, window, out[f"{nm}_w{window}"], flush=True)
    json.dump(out, open("gpurun_out/quick_bench.json", "w"), indent=1)
