"""End-to-end timing of the headless sequential odometry loop (como_amd/odom/sequential.py) on a rendered 640x480 sequence
with the parameters of the reference's config/como.yml (9 keyframes, 24 one-way frames, 64 inducing points, network input
192x256, sub-selection window 4; tracking float32, mapping float64 system with float32 pixel kernels).

    python scripts/gpu_odometry_bench.py [--frames 80] [--H 480 --W 640]

Prints one JSON line: frames/s of the whole loop and the mean cost of its parts (tracked frame, mapping iteration, keyframe
insertion, one-way insertion, tracker reference refresh).  Run on the GPU box through gpurun.
"""
import argparse
import json
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from como_amd import synth  # noqa: E402
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule  # noqa: E402
from como_amd.odom.sequential import ComoSeq  # noqa: E402


def cfgs(dev, args):
    tracking = {"device": dev, "dtype": "float", "color": "gray",
                "pyr": {"start_level": 0, "end_level": 3, "depth_interp_mode": "nearest_neighbor"},
                "term_criteria": {"max_iter": 50, "delta_norm": 1.0e-3, "rel_tol": 1.0e-3, "grad_norm": 1.0},
                "sigmas": {"photo": 1.0e-1},
                "keyframing": {"kf_depth_motion_ratio": 0.12, "kf_num_pixels_frac": 0.75, "one_way_freq": 3}}
    mapping = {"device": dev, "dtype": "double", "pix_dtype": args.pix, "color": "gray", "track_ref": {"num_keyframes": 1}, "viewer_snapshots": False,
               "graph": {"num_keyframes": 9, "num_one_way_frames": 24}, "network_size": [192, 256],
               "photo_construction": {"nonmax_suppression_window": 4, "pairwise_batch_size": 128, "radius_thresh": 0.0,
                                      "degrees_thresh": 0.0},
               "term_criteria": {"max_iter": 20, "delta_norm": 1.0e-8, "abs_tol": 1.0e-6, "rel_tol": 1.0e-6},
               "sigmas": {"photo": 1.0e-1, "mean_depth_prior": 1.0e-2, "scale_prior": 1.0e-4, "pose_prior": 1.0e-6},
               "sampling": {"mode": "greedy_conditional_entropy", "max_num_coords": 64, "max_stdev_thresh": 1.0e-2, "border": 3,
                            "fixed_var": 0.0, "dist_thresh": 1.0e-1},
               "corr": {"corr_mode": "logz", "corr_thresh": 3.0e-2, "distill_with_prior": True, "min_obs_depth": 0.0,
                        "logz_grad_mag_thresh": 7.0e-2},
               "init": {"start_level": 0, "end_level": 3, "max_iter": 50, "delta_norm": 1.0e-4, "rel_tol": 1.0e-4,
                        "kf_depth_motion_ratio": 0.04, "kf_num_pixels_frac": 0.75}}
    return {"tracking": tracking, "mapping": mapping}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--step", type=float, default=0.01)
    ap.add_argument("--pix", default="float")
    ap.add_argument("--save-traj", default=None, help="write the tracked trajectory in TUM format")
    ap.add_argument("--cprofile-after", type=int, default=-1, help="cProfile the loop from this frame on (host-side breakdown)")
    ap.add_argument("--census-after", type=int, default=-1,
                    help="count torch API calls per como_amd source line from this frame on (TorchFunctionMode)")
    ap.add_argument("--torch-profile-after", type=int, default=-1,
                    help="torch.profiler (with python stacks) from this frame on: which source lines launch the small torch kernels")
    args = ap.parse_args()
    dev = "cuda:0"
    H, W = args.H, args.W
    # the sequence pinned against the reference's own loop (scripts/ate_sequence.py SEQ640 = tests/golden/ate_sequence_640.npz)
    from scripts.ate_sequence import SEQ640, loop_cfgs, render_frames
    G = dict(SEQ640, seed=1, nframes=args.frames, H=H, W=W)
    K, T, rgbs_cpu = render_frames(G)
    rgbs = [r.to(dev) for r in rgbs_cpu]
    model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
    odo = ComoSeq(loop_cfgs(G, args.pix, dev, graph_network=True), K.clone(), (H, W), model)

    # instrument the parts (synchronising timers: this is a breakdown, the loop total below is measured without them)
    parts = {}

    def timed(obj, name, label):
        fn = getattr(obj, name)

        def wrap(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            parts.setdefault(label, []).append(time.perf_counter() - t0)
            return r
        setattr(obj, name, wrap)

    def host_timed(obj, name, label):
        # host time spent INSIDE the call (no synchronisation added): where the Python side of a frame goes
        fn = getattr(obj, name)

        def wrap(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            parts.setdefault("host:" + label, []).append(time.perf_counter() - t0)
            return r
        setattr(obj, name, wrap)

    if os.environ.get("COMO_ODO_BREAKDOWN", "1") == "2":
        from como_amd.odom import window_ba as _wba
        host_timed(odo.tracking, "handle_frame", "track_frame (incl. its read-back wait)")
        host_timed(odo.tracking, "update_kf_reference", "tracker_reference_refresh")
        host_timed(odo.mapping, "handle_tracking_data", "handle_tracking_data")
        host_timed(odo.mapping, "iterate", "mapping_iterate (retarget / rebuild + enqueue + publish)")
        host_timed(odo.mapping, "get_kf_ref_data", "get_kf_ref_data")
        host_timed(odo.mapping, "_window_state", "_window_state")
        host_timed(_wba.WindowBA, "retarget", "WindowBA.retarget")
        host_timed(_wba.WindowBA, "step", "WindowBA.step (enqueue one iteration)")
        host_timed(_wba.WindowBA, "snapshot_state", "WindowBA.snapshot_state")
    if os.environ.get("COMO_ODO_BREAKDOWN", "1") == "1":
        timed(odo.tracking, "handle_frame", "track_frame")
        timed(odo.tracking, "update_kf_reference", "tracker_reference_refresh")
        timed(odo.mapping, "iterate", "mapping_iterate")
        timed(odo.mapping, "add_keyframe", "add_keyframe")
        timed(odo.mapping, "add_one_way_frame", "add_one_way_frame")
        timed(odo.mapping, "attempt_two_frame_init", "two_frame_init_attempt")
    kinds = []
    t_frame = []                                             # (no extra synchronisation: every tracked frame reads its result back)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_first_tracked = None
    prof = None
    tprof = None
    census = None
    for k in range(args.frames):
        if k == args.cprofile_after:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        if k == args.census_after:
            from torch.overrides import TorchFunctionMode

            class Census(TorchFunctionMode):
                def __init__(self):
                    super().__init__()
                    self.n = {}

                def __torch_function__(self, func, types, a=(), kw=None):
                    f = sys._getframe(1)
                    while f is not None and "como_amd" not in f.f_code.co_filename:
                        f = f.f_back
                    if f is not None:
                        name = getattr(func, "__name__", str(func))
                        key = (f.f_code.co_filename.split("como_amd/")[-1], f.f_lineno, name)
                        self.n[key] = self.n.get(key, 0) + 1
                    return func(*a, **(kw or {}))
            census = Census()
            census.__enter__()
        if k == args.torch_profile_after:
            from torch.profiler import ProfilerActivity, profile
            tprof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
            tprof.__enter__()
        t_frame.append(time.perf_counter())
        kinds.append(odo.iter(1.0 + k, rgbs[k]))
        if t_first_tracked is None and odo.mapping.is_init:
            torch.cuda.synchronize()
            t_first_tracked = (k, time.perf_counter())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    band = None
    try:                                                   # the band median's candidate share over the window's life (csrc/densify.hip)
        ba = odo.mapping._ba or getattr(odo.mapping, "_ba_prev", None)
        for e in ba.w["dr_ws"].values():
            st = e.get("band") if isinstance(e, dict) else None
            if st is not None:
                rows = int(st["lref"].shape[1])
                band = {"calls": int(st["calls"]), "candidates_per_call_and_keyframe": float(st["ncand"].double().mean()) / max(1, int(st["calls"])),
                        "rows": rows}
    except Exception:   # noqa: BLE001
        pass
    if census is not None:
        census.__exit__(None, None, None)
        os.makedirs("gpurun_out", exist_ok=True)
        skip = {"__get__", "size", "dim", "shape", "is_contiguous", "data_ptr", "stride", "numel", "view", "reshape", "__getitem__",
                "expand", "unsqueeze", "squeeze", "permute", "transpose", "is_cuda", "device", "dtype", "element_size"}
        byline = {}
        for (fn, ln, name), c in census.n.items():
            if name in skip:
                continue
            byline.setdefault((fn, ln), []).append((name, c))
        with open("gpurun_out/odo_census.txt", "w") as f:
            nfr = args.frames - args.census_after
            f.write(f"torch API calls per source line over {nfr} frames (views / metadata excluded)\n")
            for (fn, ln), v in sorted(byline.items(), key=lambda kv: -sum(c for _, c in kv[1]))[:140]:
                f.write(f"{sum(c for _, c in v) / nfr:8.2f}/frame  {fn}:{ln}  " + ", ".join(f"{n} x{c}" for n, c in sorted(v, key=lambda t: -t[1])) + "\n")
    if tprof is not None:
        tprof.__exit__(None, None, None)
        os.makedirs("gpurun_out", exist_ok=True)
        ka = tprof.key_averages(group_by_stack_n=12)
        rows = []
        for e in ka:
            src = [f for f in e.stack if "como_amd" in f]
            rows.append((e.count, e.self_device_time_total if hasattr(e, "self_device_time_total") else e.self_cuda_time_total,
                         e.key, " <- ".join(x.strip()[-70:] for x in src[:3])))
        with open("gpurun_out/odo_torchprof.txt", "w") as f:
            f.write("count  self_device_us  op  <- como_amd frames (innermost first)\n")
            for c, t, kname, src in sorted(rows, key=lambda r: -r[0])[:150]:
                f.write(f"{c:6d} {t:12.0f}  {kname[:40]:40s} {src}\n")
    if prof is not None:
        import io
        import pstats
        prof.disable()
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("cumulative").print_stats(60)
        buf.write("\n==== by own time ====\n")
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(90)
        buf.write("\n==== window rebuild: callees of WindowBA.__init__ and below ====\n")
        st_ = pstats.Stats(prof, stream=buf).sort_stats("cumulative")
        for fn_ in ("__init__", "_prepare_topology", "_finish_topology", "_build_pair_table", "_prepare_fused", "add_one_way_frame", "_window_state"):
            st_.print_callees("window_ba.py.*" + fn_ if fn_.startswith("_") or fn_ == "__init__" else fn_)
        buf.write("\n==== who asks os.environ / converts / allocates ====\n")
        for pat in ("os.py.*__getitem__", "method 'to' of", "built-in method torch.empty", "method 'tolist'", "method 'copy_'", "built-in method torch.zeros"):
            st_.print_callers(pat)
        os.makedirs("gpurun_out", exist_ok=True)
        open("gpurun_out/odo_cprofile.txt", "w").write(buf.getvalue())
    n_tracked = args.frames - 1 - t_first_tracked[0]
    # trajectory error against GT after a similarity alignment of the translations (monocular: scale is a gauge)
    est = torch.stack([p[0, :3, 3].double().cpu() for p in odo.est_poses])
    gt = T[args.frames - len(odo.est_poses):, :3, 3]
    s = (est * gt).sum() / (est * est).sum() if (est * est).sum() > 0 else 1.0
    out = {"frames": args.frames, "size": [H, W], "loop_fps_after_init": n_tracked / (t1 - t_first_tracked[1]),
           "loop_ms_per_frame_after_init": 1e3 * (t1 - t_first_tracked[1]) / max(n_tracked, 1),
           "init_done_at_frame": t_first_tracked[0],
           "requests": {str(k): kinds.count(k) for k in set(kinds)},
           "keyframes": len(odo.mapping.kf_timestamps), "one_way_frames": len(odo.mapping.recent_timestamps),
           "landmarks": int(odo.mapping.P_m.shape[0]), "window_full": bool(odo.mapping.window_full),
           "band_median": band,
           "parts_ms": {k: {"mean": 1e3 * sum(v) / len(v), "max": 1e3 * max(v), "n": len(v)} for k, v in parts.items()},
           "traj_scale": float(s), "traj_rmse_after_scale": float(((s * est - gt) ** 2).sum(1).mean().sqrt())}
    # wall time from the start of a frame to the start of the next, by what the frame asked the mapper for (the tracker's read-back
    # of frame k + 1 waits for the mapping work of frame k: the cost of a request shows up in the frame that issued it and the next)
    t_frame.append(t1)
    by_kind = {}
    for k in range(t_first_tracked[0] + 1, args.frames):
        by_kind.setdefault(str(kinds[k]), []).append(1e3 * (t_frame[k + 1] - t_frame[k]))
    out["frame_ms_by_request"] = {k: {"mean": sum(v) / len(v), "median": sorted(v)[len(v) // 2], "max": max(v), "n": len(v)} for k, v in by_kind.items()}
    slow = sorted(((1e3 * (t_frame[k + 1] - t_frame[k]), k, str(kinds[k])) for k in range(t_first_tracked[0] + 1, args.frames)), reverse=True)[:5]
    out["slowest_frames_ms"] = [(round(t, 2), k, kd) for t, k, kd in slow]
    if args.save_traj:
        from como_amd.utils.io import save_traj
        save_traj(args.save_traj, odo.timestamps, torch.cat([p.double().cpu() for p in odo.est_poses]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
