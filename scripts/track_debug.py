"""GPU box: per-iteration record of the persistent level kernel vs the chain (sigma, nvalid, mse, |H|, |g|)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import como_amd.odom.frontend.photo_tracking as pt
from tests.test_gpu_r2 import _tracking_level_inputs
H, W = 480, 640
tp, K, P, vals, J = _tracking_level_inputs(H, W, 3)
aff = torch.zeros((1, 2, 1), device="cuda:0")
for k in range(1, 7):
    term = {"max_iter": k, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}
    pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J.clone(), tp["img_cur"], K, 0.1, term, fused=True)
    f = pt.photo_level_tracking.last_out.cpu().double()
    # chain: iterate k times manually
    T, a = tp["Tji_init"].clone(), aff.clone()
    Jc = J.clone()
    for _ in range(k):
        out, _, _, _ = pt.tracking_iter_raw(T, P, K, tp["img_cur"], a, vals, Jc, want_proj=False)
        T, a = out[80:96].reshape(1, 4, 4).clone(), out[96:98].reshape(1, 2, 1).clone()
    c = out.cpu().double()
    print(f"iter {k}: sigma f {f[101]:.9e} c {c[101]:.9e} | nv {int(f[102])} {int(c[102])} | mse {f[98]:.9e} {c[98]:.9e} | "
          f"H00 {f[0]:.9e} {c[0]:.9e} | g0 {f[64]:.6e} {c[64]:.6e} | dT {float((f[80:96]-c[80:96]).abs().max()):.2e} status {int(f[104])}")
