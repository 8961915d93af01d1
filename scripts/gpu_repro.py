import sys, torch
sys.path.insert(0, ".")
import como_amd.odom.backend.photo as photo
dev = "cuda:0"
def run(dtype, m, Hh, Ww, window, variant):
    photo.BLOCK_VARIANT = variant
    torch.manual_seed(0)
    B = 3
    n = (Hh // window) * (Ww // window)
    K = torch.tensor([[52.5, 0, (Ww-1)/2], [0, 52.5, (Hh-1)/2], [0, 0, 1]], dtype=dtype, device=dev)
    v, u = torch.meshgrid(torch.arange(0, Hh, window, dtype=dtype, device=dev), torch.arange(0, Ww, window, dtype=dtype, device=dev), indexing="ij")
    z = 1.0 + 0.1 * torch.rand((B, n), dtype=dtype, device=dev)
    ray = torch.stack(((u.reshape(-1) - K[0,2]) / K[0,0], (v.reshape(-1) - K[1,2]) / K[1,1], torch.ones(n, dtype=dtype, device=dev)), -1)
    poses = torch.eye(4, dtype=dtype, device=dev).repeat(B, 1, 1)
    poses[:, 0, 3] = 0.02 * torch.arange(B, dtype=dtype, device=dev)
    Pwn = ((z[..., None] * ray[None]) + poses[:, None, :3, 3]).permute(0, 2, 1).contiguous()
    vals = torch.rand((B, n), dtype=dtype, device=dev)
    dT = torch.randn((B, 18, n), dtype=dtype, device=dev)
    uvec = torch.randn((B, 3, n), dtype=dtype, device=dev)
    Kt = torch.rand((B, Hh * Ww, m), dtype=dtype, device=dev) / m
    pix = ((v.reshape(-1) * Ww + u.reshape(-1)).to(torch.int32))[None].repeat(B, 1).contiguous()
    invz = torch.ones((B, m), dtype=dtype, device=dev)
    dzdP = torch.randn((B, 3), dtype=dtype, device=dev)
    img = torch.rand((B, 3, Hh, Ww), dtype=dtype, device=dev)
    aff = torch.zeros((B, 2), dtype=dtype, device=dev)
    L = 3 * m
    D = 8 * B + 3 * L
    kf_inds = torch.arange(8 * B, device=dev).reshape(B, 8)
    lm = torch.stack([torch.arange(m, device=dev) + m * b for b in range(B)])
    landmark_inds = (3 * lm.repeat_interleave(3, dim=1) + torch.arange(3, device=dev).repeat(m)[None]) + 8 * B
    ref = [0, 1, 1, 2]; tgt = [1, 2, 0, 1]
    table = photo.PairTable(ref, tgt, [False] * 4, B, kf_inds, torch.empty(0, device=dev), landmark_inds, 3 * Hh * Ww, 0, dev)
    H = torch.zeros((D, D), dtype=torch.float64, device=dev); g = torch.zeros(D, dtype=torch.float64, device=dev); e = torch.zeros((), dtype=torch.float64, device=dev)
    photo.photo_system_factored(table, poses_all=poses, aff_all=aff, Pwn=Pwn, vals=vals, dPwn_dTwc=dT, uvec=uvec, Kt=Kt, pixidx=pix, invz=invz,
                                dzdP=dzdP, img_base=img, K=K, H_img=Hh, W_img=Ww, H=H, g=g, err_out=e)
    torch.cuda.synchronize()
    return H.abs().sum().item(), g.abs().sum().item(), e.item()
for cfg in [(torch.float32, 64, 48, 64, 2), (torch.float32, 8, 48, 64, 2), (torch.float64, 64, 48, 64, 2), (torch.float64, 8, 48, 64, 2), (torch.float32, 64, 48, 64, 1)]:
    for variant in (1, 0, 2):
        print(cfg[0], "m", cfg[1], "win", cfg[4], "variant", variant, flush=True)
        print("   ", run(*cfg, variant), flush=True)
