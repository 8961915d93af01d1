"""Timings of the per-frame / per-keyframe pieces that are not in bench.py's GN-iteration metric (run on the GPU box):
tracking GN iteration at 640x480, DepthCov network + run_model, image ops, greedy sampler, prep_predictor.
    python scripts/gpu_aux_bench.py            -> one JSON line
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from como_amd import synth  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    import como_amd.odom.frontend.photo_tracking as pt
    from como_amd.utils import image_processing as ip
    from como_amd.odom.backend import sparse_map as smap
    from como_amd.depth_cov.core.DepthCovModule import DepthCovModule, run_model
    from como_amd.depth_cov.core import samplers
    from como_amd.depth_cov.core.covariance import prep_predictor
    out = {}
    # tracking GN iteration, level 0 of 640x480 (N = 307,200 reference pixels)
    tp = synth.make_tracking_pair(H=480, W=640, dtype=torch.float32, device=DEV, seed=3, levels=1)
    K = tp["intrinsics"]
    stack = ip.img_and_grads(tp["img_ref"])
    v, u = torch.meshgrid(torch.arange(480., device=DEV), torch.arange(640., device=DEV), indexing="ij")
    z = tp["depth_ref"][0, 0]
    ray = torch.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)), -1).reshape(-1, 3)
    P = (z.reshape(-1, 1) * ray)[None].contiguous()
    vals = tp["img_ref"].reshape(1, -1, 1).contiguous()
    dI = torch.stack((stack[0, 1].reshape(-1), stack[0, 2].reshape(-1)), -1)[None, :, None, :].contiguous()
    J = pt.precalc_jacobians(dI, P, vals, K)
    aff = torch.zeros((1, 2, 1), device=DEV)
    out["tracking_iter_640x480_ms"] = timeit(lambda: pt.tracking_iter_raw(tp["Tji_init"], P, K, tp["img_cur"], aff, vals, J,
                                                                          want_proj=False))
    out["tracking_px_per_s"] = 307200 / (out["tracking_iter_640x480_ms"] * 1e-3)
    term = {"max_iter": 10, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}      # exactly 10 iterations
    out["tracking_level_10iters_graph_ms"] = timeit(lambda: pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J, tp["img_cur"], K,
                                                                                    0.1, term), n=5, warm=2)
    out["tracking_level_10iters_eager_ms"] = timeit(lambda: pt.photo_level_tracking(tp["Tji_init"], aff, vals, P, J, tp["img_cur"], K,
                                                                                    0.1, term, use_graph=False), n=5, warm=2)
    out["precalc_jacobians_ms"] = timeit(lambda: pt.precalc_jacobians(dI, P, vals, K))
    # image ops
    img = tp["img_ref"]
    out["img_and_grads_640x480_ms"] = timeit(lambda: ip.img_and_grads(img))
    pyr = ip.ImagePyramidModule(1, 0, 3, DEV, torch.float32)
    out["pyramid3_ms"] = timeit(lambda: pyr(img))
    out["subselect_w4_ms"] = timeit(lambda: smap.subselect_pixels(stack, 4))
    # DepthCov network (seeded weights), 192x256 network size, 480x640 image
    model = DepthCovModule(synth.depthcov_state_dict(0, device=DEV))
    rgb = torch.rand(1, 3, 480, 640, device=DEV)
    out["depthcov_run_model_ms"] = timeit(lambda: run_model(model, rgb), n=10)
    rgb_r = torch.rand(1, 3, 192, 256, device=DEV)
    out["depthcov_forward_192x256_eager_ms"] = timeit(lambda: model(rgb_r), n=10)
    out["depthcov_forward_192x256_graph_ms"] = timeit(lambda: model.forward_graphed(rgb_r), n=10)
    # greedy sampler: 64 points on the 192x256 covariance image (domain 49,152 pixels)
    cov = synth.synthetic_cov_params(1, 192, 256, seed=2, dtype=torch.float64).float().to(DEV)
    sv = torch.tensor(1.0)
    out["greedy_sampler_64pts_ms"] = timeit(lambda: samplers.sample_sparse_coords(cov, 64, "greedy_conditional_entropy", border=2,
                                                                                 dist_thresh=0.02, signal_var=sv), n=3, warm=1)
    # conditioning of one keyframe: K_mm, Cholesky, K~ (307,200 x 64), float64 as the mapping dtype
    cov64 = synth.synthetic_cov_params(1, 480, 640, seed=2, dtype=torch.float64).to(DEV)
    coords, _ = samplers.sample_sparse_coords(cov, 64, "greedy_conditional_entropy", border=2, dist_thresh=0.02, signal_var=sv)
    cm = (coords.double() * 2.5)
    out["prep_predictor_f64_ms"] = timeit(lambda: prep_predictor(cov64, cm, 1.0), n=5, warm=1)
    print(json.dumps({k: round(v, 4) for k, v in out.items()}))


if __name__ == "__main__":
    main()
