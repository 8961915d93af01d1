"""Finer breakdown of the odometry loop's per-frame host + device time (synchronising timers around the sub-steps of
Tracking.handle_frame / update_kf_reference and MappingSeq.map).  python scripts/gpu_odometry_profile.py [--frames 60]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scripts.gpu_odometry_bench as ob  # noqa: E402
from como_amd import synth  # noqa: E402
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule  # noqa: E402
import como_amd.odom.Tracking as trk_mod  # noqa: E402
import como_amd.odom.Mapping as map_mod  # noqa: E402
import como_amd.odom.sequential as seq_mod  # noqa: E402
import como_amd.odom.frontend.photo_tracking as pt  # noqa: E402

parts = {}


def timed(owner, name, label):
    fn = getattr(owner, name)

    def wrap(*a, **k):
        if torch.cuda.is_current_stream_capturing():        # (a wrapped call inside a hipGraph capture: no timers there)
            return fn(*a, **k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        parts.setdefault(label, []).append(time.perf_counter() - t0)
        return r
    setattr(owner, name, wrap)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--pix", default="float")
    ap.add_argument("--step", type=float, default=0.01)
    args = ap.parse_args()
    args.H, args.W = 480, 640
    dev = "cuda:0"
    scene = synth.PlaneScene(seed=1, freq_scale=1.0)
    K = synth.intrinsics_for(480, 640)
    T = synth.gt_poses(args.frames, step=args.step, deg=0.3)
    rgbs = [scene.render(T[k], K, 480, 640)[0][None, None].repeat(1, 3, 1, 1).to(dev) for k in range(args.frames)]
    model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
    timed(trk_mod, "photo_tracking_pyr", "trk.photo_tracking_pyr")
    timed(pt, "photo_level_tracking", "trk.level")
    timed(trk_mod, "precalc_jacobians", "ref.precalc_jacobians")
    timed(trk_mod, "fill_image", "trk.fill_image")
    timed(seq_mod, "transfer_data", "transfer_data")
    import como_amd.odom.window_ba as wba
    timed(wba.WindowBA, "_prepare_topology", "ba.rebuild.prepare_topology")
    timed(wba.WindowBA, "_prepare_fused", "ba.rebuild.prepare_fused")
    timed(wba.WindowBA, "step", "ba.step")
    timed(map_mod, "WindowBA", "ba.rebuild.total")
    timed(map_mod, "track_and_init", "kf.track_and_init")
    timed(map_mod, "_run_model", "kf.run_model")
    timed(map_mod, "_prep_predictor", "kf.prep_predictor")
    import como_amd.odom.frontend.corr as corr
    for nm in ("reproject_points", "distill_depth_from_scratch", "distill_conditional_depth_from_scratch", "sample_sparse_coords",
               "filter_reproj_coords", "_sample_at"):
        timed(corr, nm, "kf.corr." + nm)
    odo = seq_mod.ComoSeq(ob.cfgs(dev, args), K.clone(), (480, 640), model)
    timed(odo.tracking, "prep_tracking_img", "trk.prep_tracking_img")
    timed(odo.tracking, "get_reproj_last_kf", "trk.get_reproj_last_kf")
    timed(odo.tracking, "handle_frame", "trk.handle_frame")
    timed(odo.tracking, "update_kf_reference", "ref.update_kf_reference")
    timed(odo.tracking, "depth_pyr_module", "ref.depth_pyr")
    timed(odo.mapping, "get_kf_ref_data", "map.get_kf_ref_data")
    timed(odo.mapping, "iterate", "map.iterate")
    timed(odo.mapping, "handle_tracking_data", "map.handle_tracking_data")
    timed(odo.mapping, "map", "map.map")
    iters = []
    for k in range(args.frames):
        odo.iter(1.0 + 0.033 * k, rgbs[k])
        if hasattr(pt.photo_level_tracking, "last_iters"):
            iters.append(pt.photo_level_tracking.last_iters)
    med = lambda v: sorted(v)[len(v) // 2]
    out = {k: {"mean_ms": round(1e3 * sum(v) / len(v), 3), "median_ms": round(1e3 * med(v), 3), "total_ms": round(1e3 * sum(v), 1),
               "n": len(v)} for k, v in sorted(parts.items())}
    out["finest_level_iters_mean"] = sum(iters) / max(len(iters), 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
