"""bench.py -- GN iterations / second of the 8-keyframe 640x480 photometric window BA on MI355X.

    python bench.py --gpus N --steps K --warmup W [--window 1|4] [--dtype f32|f64] [--no-cpu]
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one Gauss-Newton iteration of the window BA, exactly the reference's Mapping.iterate sequence
(como/odom/Mapping.py:760-968): project landmarks -> dense reference points -> photometric normal equations of all
14 keyframe pairs (warp, residual, exact global median, Huber, Jacobian rows, J^T J / J^T r) -> priors -> dense
Cholesky solve -> pose / affine / landmark update.  Inputs (images, K~, landmarks) are resident in HBM before the
timed region.  Synthetic seeded scene (como_amd/synth.py); no datasets or checkpoints exist in this image.

Workloads: --window 1 (default) = every pixel is a reference pixel (n = 307,200 per keyframe, 4.3 M pixel-pairs per
iteration); --window 4 = the reference's default sub-selection (config/como.yml:37, n = 19,200).  The per-pixel path
runs in --dtype (f32 default, mixed precision: the normal equations, priors, solve and state are float64 always).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from como_amd import dist as cdist  # noqa: E402
from como_amd import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E peak (guides/MI355X_MICROARCH.md)
ALGO_SCALARS_PER_PIXEL_PAIR = 98  # SURVEY.md section 8(d) unit B: 34 + m scalars, m = 64


def build_state(args, device, pix_dtype):
    from como_amd.depth_cov.core.covariance import prep_predictor

    def predictor(cov, coords_m):
        Kinv, L, Kt = prep_predictor(cov.double(), coords_m.double(), 1.0)     # conditioning in f64
        return Kinv, L, Kt.to(pix_dtype)

    return synth.make_window(B=args.keyframes, H=args.height, W=args.width, m=64, dtype=torch.float64, device=device,
                             seed=args.seed, predictor=predictor)


def cpu_baseline(args, state_cpu):
    """The oracle ("port") timed on the host cores: same algorithmic structure as the reference's CPU path
    (materialised Jacobian rows, batched Gram products, index_add assembly, cholesky_ex + cholesky_solve)."""
    from oracle import dense_ref as odr, photo_ba as oba
    st = state_cpu
    K = st["intrinsics"][0]
    B = st["kf_poses"].shape[0]
    H, W = st["kf_img_and_grads"].shape[-2:]
    m = st["coords_m"].shape[1]
    Pb, ids = odr.batched_landmarks(st["P_m"], st["correspondence_mask"])
    med0 = st["median_depth_init"].double()
    L = st["P_m"].shape[0]
    D = 8 * B + 3 * L
    kf_inds = torch.arange(8 * B).reshape(B, 8)
    lm = (3 * ids.repeat_interleave(3, dim=1) + torch.arange(3).repeat(m)[None]) + 8 * B
    cn = odr.subselect_pixels(st["kf_img_and_grads"], 4)            # bounded sample: the window-4 sub-selection
    bi = torch.arange(B)[:, None].expand(-1, cn.shape[1])
    Kt_rows = st["Knm_Kmminv"][bi, cn[..., 0], cn[..., 1], :].double()
    vals = st["kf_img_and_grads"][bi, :1, cn[..., 0], cn[..., 1]]
    ref, tgt = oba.consecutive_pairs(B)
    rid, tid = torch.tensor(ref), torch.tensor(tgt)

    def one_iter():
        t0 = time.perf_counter()
        p, logz, zm, dlogz_dz, dz_dPw, dz_dTwc, dp_dPw, dp_dTwc = odr.project_landmarks(st["kf_poses"], Pb, K, Pb, med0)
        Pw, dT, dz, med, _ = odr.dense_reference(logz, st["kf_poses"], Kt_rows, cn, K, dlogz_dz @ dz_dTwc, dlogz_dz)
        Hm = torch.zeros((D, D), dtype=torch.float64)
        g = torch.zeros(D, dtype=torch.float64)
        oba.batch_photo_cost(vals[rid], st["kf_aff_params"][rid], Pw[rid], st["kf_poses"][tid], st["kf_aff_params"][tid],
                             st["kf_img_and_grads"][tid], dT[rid], dz[rid], dz_dPw[rid], kf_inds[rid], kf_inds[tid], lm[rid], K, Hm, g)
        t1 = time.perf_counter()
        Hm += 1e3 * torch.eye(D, dtype=torch.float64)               # stand-in for the O(B m^2) priors: keeps H PD for the solve
        oba.solve_system(Hm, g)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    # pick the fastest thread count for these small-tensor ops (256 threads on a 256-core host is ~100x slower than 16)
    best = None
    for nt in sorted({8, 16, 32, min(64, os.cpu_count() or 8)}):
        if nt > (os.cpu_count() or 8):
            continue
        torch.set_num_threads(nt)
        a, b2 = one_iter()
        if best is None or a + b2 < best[0]:
            best = (a + b2, nt)
    torch.set_num_threads(best[1])
    one_iter()
    reps = 3
    lin, sol = zip(*[one_iter() for _ in range(reps)])
    lin, sol = sorted(lin)[reps // 2], sorted(sol)[reps // 2]
    scale = (args.height * args.width) / cn.shape[1] if args.window == 1 else 16.0 / (args.window ** 2)
    t_iter = lin * scale + sol
    return {"value": 1.0 / t_iter, "unit": "GN iters/s", "cores": best[1], "kind": "port",
            "sample": f"oracle (torch-CPU, float64) GN iteration on the window-4 sub-selection of the same window "
                      f"(n={cn.shape[1]} px/KF, {len(ref)} pairs, D={D}), median of {reps}: linearise {lin * 1e3:.0f} ms, "
                      f"solve {sol * 1e3:.0f} ms; linearisation time scaled x{scale:g} to this workload's pixel count"}


def odometry_loop(device, frames=100):
    """Informational: the whole headless sequential odometry loop (tracking + keyframe management + one mapping iteration
    per frame, como_amd/odom/sequential.py) on a rendered 640x480 sequence with the parameters of the reference's
    config/como.yml; frames/s after the two-frame initialisation.  Never fails the bench line."""
    try:
        import argparse
        from como_amd import synth
        from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
        from como_amd.odom.sequential import ComoSeq
        from scripts.gpu_odometry_bench import cfgs
        H, W = 480, 640
        scene = synth.PlaneScene(seed=1, freq_scale=1.0, device=device)
        K = synth.intrinsics_for(H, W, device=device)
        T = synth.gt_poses(frames, step=0.01, deg=0.3, device=device)
        rgbs = [scene.render(T[k], K, H, W)[0][None, None].repeat(1, 3, 1, 1) for k in range(frames)]
        model = DepthCovModule({k: v.to(device) for k, v in synth.depthcov_state_dict(0).items()})
        odo = ComoSeq(cfgs(str(device), argparse.Namespace(pix="float")), K.cpu().clone(), (H, W), model)
        t0, k0, kinds = None, None, []
        for k in range(frames):
            kinds.append(odo.iter(1.0 + 0.033 * k, rgbs[k]))
            if t0 is None and odo.mapping.is_init:
                torch.cuda.synchronize()
                t0, k0 = time.perf_counter(), k
        torch.cuda.synchronize()
        n = frames - 1 - k0
        el = time.perf_counter() - t0
        return {"workload": "sequential odometry loop, rendered 640x480 sequence, config/como.yml parameters (9 keyframes, 24 one-way "
                            "frames, m=64, window 4, float32 tracking, float64 mapping system / float32 pixel kernels)",
                "value": n / el, "unit": "frames/s", "ms_per_frame": 1e3 * el / n, "frames": n,
                "keyframes_inserted": kinds.count("keyframe"), "one_way_inserted": kinds.count("one-way")}
    except Exception as e:                                  # noqa: BLE001
        return {"error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--window", type=int, default=1, choices=[1, 2, 4, 8])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--keyframes", type=int, default=8)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--eager", action="store_true", help="do not capture the iteration into a hipGraph")
    ap.add_argument("--no-secondary", action="store_true", help="skip the window-4 (reference default sub-selection) line")
    args = ap.parse_args()

    shard, device = cdist.init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    if shard.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={shard.world}")
    pix_dtype = torch.float32 if args.dtype == "f32" else torch.float64

    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    import copy
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = args.window
    state = build_state(args, device, pix_dtype)
    wb = WindowBA(state, cfg=cfg, pix_dtype=pix_dtype, window_full=True, shard=(shard if shard.world > 1 else None))
    pose0 = wb.kf_poses.clone()

    for _ in range(args.warmup):
        wb.iterate()
    graphed = (not args.eager) and wb.capture()
    if not graphed and not args.eager and shard.rank == 0 and shard.world == 1:
        print("hipGraph capture failed:", getattr(wb, "capture_error", "?"), file=sys.stderr)
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wb.step()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = shard.max_scalar(time.perf_counter() - t0, device)

    # roofline probe: the dominant kernel's duration from HIP events on the launch stream (events cannot sit inside a
    # captured graph, so the same kernel on the same data is timed in eager iterations right after the timed region;
    # profiles/ holds the rocprofv3 --kernel-trace --stats summary of this command for cross-checking)
    wb.events = {}
    for _ in range(5):
        wb.iterate()
    torch.cuda.synchronize()
    ms_step = elapsed / args.steps * 1e3
    ev = wb.events.get("blocks", [])
    blk_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    npairs = wb.table.b
    pb, pe = wb.pix_range if wb.pix_range is not None else (0, wb.n)
    pixel_pairs_rank = npairs * (pe - pb)
    bytes_per = ALGO_SCALARS_PER_PIXEL_PAIR * (4 if args.dtype == "f32" else 8)
    achieved = pixel_pairs_rank * bytes_per / (blk_ms * 1e-3) / 1e9 if blk_ms > 0 else 0.0
    info = int(__import__("como_amd.odom.backend.linear_system", fromlist=["x"]).solve_system.last_info)
    pose_err = (wb.kf_poses - state["poses_gt"]).abs().max().item()
    pose_err0 = (pose0 - state["poses_gt"]).abs().max().item()

    # HBM traffic of the dominant kernel: rocprofv3 PMC passes of THIS command, committed under profiles/
    # (scripts/collect_profiles.sh); FETCH_SIZE doubled per the gfx950 note of guides/MI355X_MICROARCH.md, WRITE_SIZE as is.
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r1_bench_pmc_summary.json")
    if os.path.exists(pmc_path) and args.gpus == 1 and args.window == 1 and args.dtype == "f32":
        try:
            pm = json.load(open(pmc_path))
            key = [k for k in pm if "ba_blocks" in k][0]
            traffic = (2.0 * pm[key]["FETCH_SIZE"]["avg_per_launch"] + pm[key]["WRITE_SIZE"]["avg_per_launch"]) * 1024.0
        except Exception:                                   # noqa: BLE001  (a malformed summary just leaves traffic null)
            traffic = None

    secondary = None
    if args.window == 1 and not args.no_secondary and args.gpus == 1:
        # the reference's default sub-selection (config/como.yml: nonmax_suppression_window 4, n = 19,200 px / keyframe)
        cfg4 = copy.deepcopy(DEFAULT_CFG)
        cfg4["photo_construction"]["nonmax_suppression_window"] = 4
        wb4 = WindowBA(build_state(args, device, pix_dtype), cfg=cfg4, pix_dtype=pix_dtype, window_full=True)
        for _ in range(args.warmup):
            wb4.iterate()
        g4 = (not args.eager) and wb4.capture()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(args.steps):
            wb4.step()
        torch.cuda.synchronize()
        e4 = time.perf_counter() - t4
        secondary = {"workload": f"same window, nonmax_suppression_window=4 (n={wb4.n} reference px/KF, the reference's default)",
                     "value": args.steps / e4, "unit": "GN iters/s", "ms_per_step": e4 / args.steps * 1e3, "hip_graph": bool(g4)}

    odometry = None
    if args.window == 1 and not args.no_secondary and args.gpus == 1:
        odometry = odometry_loop(device)

    if shard.rank == 0:
        out = {
            "metric": "GN iters/sec, 8-keyframe 640x480 photometric BA",
            "value": args.steps / elapsed, "unit": "GN iters/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.keyframes}-keyframe {args.width}x{args.height} window BA, {npairs} keyframe pairs, "
                                   f"n={wb.n} reference px/KF (window={args.window}), m=64, D={wb.dim}; one step = full GN "
                                   f"iteration (scaffold, dense ref, photometric system, priors, Cholesky solve, update)",
                       "pixel_pairs_per_iter": npairs * wb.n, "system_dim": wb.dim, "pix_dtype": args.dtype,
                       "system_dtype": "f64", "hip_graph": bool(graphed), "parallelism": f"dp{args.gpus} (reference-pixel shards of every pair)"},
            "roofline": {"bound": "hbm", "kernel": "ba_blocks_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel_ms": blk_ms,
                         "algorithmic_bytes_per_launch": pixel_pairs_rank * bytes_per},
            "solution": {"cholesky_info": info, "max_pose_abs_err_vs_gt_start": pose_err0, "max_pose_abs_err_vs_gt_end": pose_err},
        }
        if secondary is not None:
            out["secondary"] = secondary
        if odometry is not None:
            out["odometry_loop"] = odometry
        if not args.no_cpu and args.gpus == 1:
            st_cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in state.items()}
            st_cpu["Knm_Kmminv"] = st_cpu["Knm_Kmminv"]
            out["cpu_baseline"] = cpu_baseline(args, st_cpu)
        print(json.dumps(out))
    shard.barrier()


if __name__ == "__main__":
    main()
