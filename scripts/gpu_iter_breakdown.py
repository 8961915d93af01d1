"""Wall-clock breakdown of one window-BA GN iteration (sections separated by device syncs)."""
import copy
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from como_amd import synth  # noqa: E402
from como_amd.depth_cov.core.covariance import prep_predictor  # noqa: E402
from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG  # noqa: E402
import como_amd.odom.backend.linear_system as lin_sys  # noqa: E402

dev = "cuda:0"
window = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def pred(cov, cm):
    a, b, c = prep_predictor(cov.double(), cm.double(), 1.0)
    return a, b, c.float()


st = synth.make_window(B=8, H=480, W=640, m=64, dtype=torch.float64, device=dev, seed=0, predictor=pred)
cfg = copy.deepcopy(DEFAULT_CFG)
cfg["photo_construction"]["nonmax_suppression_window"] = window
wb = WindowBA(st, cfg=cfg, pix_dtype=torch.float32)
for _ in range(3):
    wb.iterate()
torch.cuda.synchronize()


def timed(fn, n=10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, out


res = {}
res["linearize_total_ms"], _ = timed(wb.linearize)
H, g = wb.linearize()
res["solve_ms"], delta = timed(lambda: lin_sys.solve_system(H, g))
res["update_ms"], _ = timed(lambda: lin_sys.update_vars(delta, wb.kf_poses, wb.kf_aff_params, wb.kf_inds, wb.recent_poses,
                                                        wb.recent_aff_params, wb.recent_inds, wb.P_m, wb.lm_start))
res["iterate_ms"], _ = timed(wb.iterate)
# CPU-side launch cost only (no sync inside): how long the host needs to ENQUEUE one iteration
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    wb.iterate()
res["host_enqueue_ms"] = (time.perf_counter() - t) / 10 * 1e3
torch.cuda.synchronize()
print(json.dumps(res, indent=1))
