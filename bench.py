"""bench.py -- GN iterations / second of the 8-keyframe 640x480 photometric window BA on MI355X.

    python bench.py --gpus N --steps K --warmup W [--window 1|4] [--dtype f64|f32] [--keyframes B] [--replicas] [--no-cpu]
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`; a plain
     `python bench.py --gpus N` with no WORLD_SIZE in the environment starts its N ranks itself, see self_launch)

One "step" = one Gauss-Newton iteration of the window BA, exactly the reference's Mapping.iterate sequence
(como/odom/Mapping.py:760-968): project landmarks -> dense reference points -> photometric normal equations of all
14 keyframe pairs (warp, residual, exact global median, Huber, Jacobian rows, J^T J / J^T r) -> priors -> dense
Cholesky solve -> pose / affine / landmark update.  Inputs (images, K~, landmarks) are resident in HBM before the
timed region.  Synthetic seeded scene (como_amd/synth.py); no datasets or checkpoints exist in this image.

Headline workload: --window 1 = every pixel is a reference pixel (n = 307,200 per keyframe, 4.3 M pixel-pairs per
iteration), per-pixel path in --dtype: **f64 by default = the reference's mapping dtype** (config/como.yml:28 `dtype: double`);
the normal equations / priors / solve / state are float64 always.  `value`, `dtype`, `roofline` (784 B per pixel-pair) and the
top-level `cpu_baseline` (the oracle on the SAME dense workload) all describe that leg.  The same JSON line also carries (N = 1):
  mixed_precision : the same dense window with the per-pixel path in float32 (392 B per pixel-pair), own roofline;
  secondary       : the window at the reference's default sub-selection (nonmax_suppression_window 4, n = 19,200), f64 and f32
                    -- the reference's operating point; its numbers are repeated as flat keys of `config` (window4_*);
  tracking        : config 2 -- the 2-frame 640x480 tracking GN iteration (unit A: 53 B per pixel);
  odometry_loop   : the whole headless sequential loop (frames / s);
  ate_vs_ref      : ATE-RMSE of the HIP loop's trajectory against the reference's own on a 72-frame 192x256 sequence;
  cpu_baseline    : the oracle (CPU restatement of the reference algorithm) timed on this box's host cores: the dense headline
                    workload (bounded: two iterations) and, nested under `window4`, the window-4 workload.
--keyframes 32 runs config 4's window (62 pairs, D ~ 2.9 k).
--replicas runs config 5's mode: ONE SEQUENCE PER GPU -- every rank drives its own rendered 640x480 sequence through the whole
odometry loop (tracking + keyframe management + DepthCov network + window BA, como_amd/odom/sequential.py <-> the reference's
como/odom/sequential/ComoSeq.py:42-127), no collective; K = frames timed per rank after the two-frame initialisation and W
warm-up frames; value = sum over ranks of frames / s ("scaling": "weak").
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL / tensor sharing fail with the legacy mode)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

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


def cpu_baseline(args, state_cpu, reps=5):
    """The oracle ("port": oracle/window.py OracleWindow.iterate = the reference's Mapping.iterate sequence with
    materialised Jacobian rows, batched Gram products, index_add assembly, the real prior factors, cholesky_ex +
    cholesky_solve) timed on the host cores -- no scaling, no stand-ins.  Top level = the SAME workload as `value` (the dense
    window, bounded to two iterations: ~10 s each and ~30 GB of materialised Jacobians); `window4` = the reference's default
    sub-selection (compare with `secondary`)."""
    from oracle.window import OracleWindow
    ncpu = os.cpu_count() or 8
    # these are many small/medium tensor ops: torch with every core of a 100+ core host is far slower than with 16-32
    best = None
    for nt in sorted({8, 16, 32, min(64, ncpu)}):
        if nt > ncpu:
            continue
        torch.set_num_threads(nt)
        ow = OracleWindow(state_cpu, window=4)
        t0 = time.perf_counter()
        ow.iterate()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    ow = OracleWindow(state_cpu, window=4)
    ow.iterate()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ow.iterate()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    n = ow.cn.shape[1]
    w4 = {"value": 1.0 / med, "unit": "GN iters/s", "cores": best[1], "host_cores": ncpu, "kind": "port", "workload": "window=4",
          "protocol": f"median of {reps} after 1 warm-up; threads = fastest of 8/16/32/64 in a one-iteration probe",
          "sample": f"oracle/window.py (torch-CPU, float64) full GN iteration incl. priors + solve on the window-4 workload "
                    f"exactly (n={n} px/KF, {ow.aux['valid'].shape[0]} pairs, D={ow.D}): median of {reps} after 1 warm-up = "
                    f"{med * 1e3:.0f} ms (min {ts[0] * 1e3:.0f}, max {ts[-1] * 1e3:.0f}); {best[1]} torch threads (fastest of "
                    f"8/16/32/64) on a {ncpu}-core host; compare with `secondary` (same workload on the GPU)"}
    del ow
    # the headline (dense) workload, not extrapolated: only where the host has the memory for it
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:                                       # noqa: BLE001
        avail = 0
    dense_ok = args.window == 1 and not args.no_cpu_dense and avail > 96e9
    if not dense_ok:
        out = dict(w4)
        out["note"] = ("the dense (window 1) oracle iteration was skipped (host memory / --no-cpu-dense / --window): this is the "
                       "window-4 workload -- compare with `secondary`, not with `value`")
        out["window4"] = w4
        return out
    torch.set_num_threads(best[1])
    owd = OracleWindow(state_cpu, window=args.window)
    td = []
    for _ in range(2):
        t0 = time.perf_counter()
        owd.iterate()
        td.append(time.perf_counter() - t0)
    nd, pairs, D = owd.cn.shape[1], owd.aux["valid"].shape[0], owd.D
    del owd
    return {"value": 1.0 / min(td), "unit": "GN iters/s", "cores": best[1], "host_cores": ncpu, "kind": "port",
            "protocol": "best of 2 (no warm-up; ~8-10 s and ~30 GB of materialised Jacobians per iteration keep the default run "
                        "within minutes -- SURVEY 8(d) asks for the median of >= 5 after 2 warm-ups at os.cpu_count() threads: "
                        "`window4` below follows the median-of-5 part; threads are the fastest of 8/16/32/64 probed on window 4)",
            "workload": f"window={args.window} (the workload of `value`)",
            "sample": f"oracle/window.py (torch-CPU, float64) full GN iteration incl. priors + solve on the headline workload exactly "
                      f"(n={nd} px/KF, {pairs} pairs, D={D}): two iterations, {td[0]:.1f} s and {td[1]:.1f} s, value = 1 / the faster; "
                      f"{best[1]} torch threads (fastest of 8/16/32/64 on the window-4 workload) on a {ncpu}-core host",
            "seconds": td, "window4": w4}


def odometry_loop(device, frames=100, seed=1, pix="float", warm=0, barrier=None, timed_frames=None):
    """The whole headless sequential odometry loop (tracking + keyframe management + DepthCov network on every new keyframe +
    one mapping iteration per frame, como_amd/odom/sequential.py <-> como/odom/sequential/ComoSeq.py:42-127) on the rendered
    640x480 sequence that is PINNED against the reference's own loop (scripts/ate_sequence.py SEQ640; seed 1, 100 frames =
    tests/golden/ate_sequence_640.npz, checked by tests/test_gpu_r5.py incl. a perturbed-network run) with the parameters of the
    reference's config/como.yml; frames/s after the two-frame initialisation (+ `warm` further untimed frames).  `seed` picks the
    scene texture and the noise (--replicas: one sequence per rank); pix = "float" (mixed precision) / "double" (the reference's
    mapping dtype) per-pixel kernels of the window BA.  timed_frames: time EXACTLY that many frames (the sequence is rendered long
    enough); barrier: called right before / after the timed region (multi-rank).  Reported beside the rate: the frame at which the
    initialisation completed, keyframes / one-way frames inserted, the trajectory error against the ground truth (scale-aligned:
    monocular) and, for the pinned seed, decisions and ATE against the reference's trajectory.
    Never fails the bench line: returns {"error": ...} instead."""
    try:
        from como_amd import synth
        from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
        from como_amd.odom.sequential import ComoSeq
        from como_amd.utils.ate import ate_rmse
        from scripts.ate_sequence import KIND_CODE, SEQ640, loop_cfgs, render_frames
        if timed_frames is not None:
            frames = timed_frames + warm + 8                # the two-frame initialisation completes on the third frame
        G = dict(SEQ640, seed=seed, nframes=frames)
        K, T, rgbs_cpu = render_frames(G)
        rgbs = [r.to(device) for r in rgbs_cpu]
        model = DepthCovModule({k: v.to(device) for k, v in synth.depthcov_state_dict(0).items()})
        odo = ComoSeq(loop_cfgs(G, pix, str(device), graph_network=True), K.clone(), (G["H"], G["W"]), model)
        from como_amd.odom import window_ba as _wba
        import gc
        gc.collect()                                        # (what an earlier leg / pass left behind is not this loop's garbage: a
        spec0 = dict(_wba.SPEC_STATS)                       #  generation-2 collection of it cost one frame of the pass 4 ms)
        t0, k0, kinds, k_end = None, None, [], frames
        t_start = []
        k_init = None
        poses = {}
        for k in range(frames):
            if k_init is None and odo.mapping.is_init:
                k_init = k
            if t0 is None and k_init is not None and k >= k_init + warm:
                if barrier is not None:
                    barrier()
                torch.cuda.synchronize()
                t0, k0 = time.perf_counter(), k
                if timed_frames is not None:
                    k_end = k + timed_frames
            if k >= k_end:
                break
            nb = len(odo.est_poses)
            t_start.append(time.perf_counter())              # (no synchronisation added: a tracked frame reads its result back)
            kinds.append(odo.iter(1.0 + k, rgbs[k]))
            if len(odo.est_poses) > nb:
                poses[k] = odo.est_poses[-1]                # (device tensors: read back after the timed region)
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        el = t_end - t0
        if barrier is not None:
            barrier()
        n = min(k_end, frames) - k0
        # wall time from the start of a frame to the start of the next, by what the frame asked the mapper for (None = a plain
        # tracked frame): the tracker's read-back of the next frame waits for the mapping work this frame queued
        t_start.append(t_end)
        by_req = {}
        for j in range(k0, len(kinds)):
            by_req.setdefault(str(kinds[j]), []).append(1e3 * (t_start[j + 1] - t_start[j]))
        frame_ms = {r: {"mean_ms": sum(v) / len(v), "median_ms": sorted(v)[len(v) // 2], "frames": len(v)} for r, v in by_req.items()}
        if timed_frames is not None and n != timed_frames:
            return {"error": f"only {n} of {timed_frames} frames could be timed (initialisation at frame {k_init})"}
        from como_amd.odom.frontend.photo_tracking import photo_tracking_pyr
        tracked = sorted(poses)
        est = [poses[k].detach().cpu().double().reshape(4, 4) for k in tracked]
        out = {"workload": f"sequential odometry loop, rendered 640x480 sequence (scripts/ate_sequence.py SEQ640, seed {seed}), config/como.yml "
                           f"parameters (9 keyframes, 24 one-way frames, m=64, window 4, float32 tracking, float64 mapping system / {pix} "
                           "pixel kernels)",
               "value": n / el, "unit": "frames/s", "ms_per_frame": 1e3 * el / n, "frames": n, "elapsed_s": el,
               "init_completed_at_frame": k_init, "keyframes_inserted": kinds.count("keyframe"), "one_way_inserted": kinds.count("one-way"),
               "tracking_chain_fallbacks": int(getattr(photo_tracking_pyr, "fallbacks", 0)), "frame_ms_by_request": frame_ms,
               # pair tables of the next one-way topology built while the tracker ran / adopted by the re-target that followed
               "pair_tables_built_ahead": {k: v - spec0.get(k, 0) for k, v in _wba.SPEC_STATS.items() if v - spec0.get(k, 0)},
               "solver_fallbacks": int(getattr(odo.mapping, "solver_fallbacks", 0)),
               "ate_vs_gt_sim3_m": float(ate_rmse(est, [T[k] for k in tracked], "sim3")) if len(est) > 3 else None}
        ref_path = os.path.join(ROOT, "tests", "golden", "ate_sequence_640.npz")
        if seed == 1 and os.path.exists(ref_path):          # the pinned sequence: the reference's own loop on the same frames
            R = np.load(ref_path)
            m = min(len(kinds), int(R["nframes"]))
            codes = [KIND_CODE[x] for x in kinds[:m]]
            both = [k for k in tracked if k < m and bool(R["tracked"][k])]
            out["vs_reference_loop"] = {
                "frames_compared": m, "same_decisions": int(sum(int(a == int(b)) for a, b in zip(codes, R["kinds"][:m]))),
                "ate_rmse_m": float(ate_rmse([poses[k].detach().cpu().double().reshape(4, 4) for k in both],
                                             [torch.from_numpy(R["T_w_curr"][k]) for k in both])) if len(both) > 3 else None,
                "fixture": "tests/golden/ate_sequence_640.npz (reference ComoSeq loop, tests/golden/make_golden_r2.py ate640)"}
        return out
    except Exception as e:                                  # noqa: BLE001
        return {"error": repr(e)[:300]}


ALGO_SCALARS_PER_TRACKING_PIXEL = 53.0 / 4.0   # SURVEY.md section 8(d) unit A: 53 B per pixel-iteration in float32


def git_sha():
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:                                       # noqa: BLE001  (the GPU box snapshot has no .git)
        return None


def committed_traffic(kernel_substr, summaries=("r6_bench_pmc_summary.json", "r5_bench_pmc_summary.json", "r4_bench_pmc_summary.json", "r3_bench_pmc_summary.json")):
    """HBM bytes per launch of a kernel from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled per the
    gfx950 note of guides/MI355X_MICROARCH.md, WRITE_SIZE as is).  NOT measured by this run: counters need rocprofv3
    around the process (scripts/collect_profiles.sh); the value is labelled with its source file."""
    for summary in summaries:
        path = os.path.join(ROOT, "profiles", summary)
        try:
            pm = json.load(open(path))
            key = [k for k in pm if kernel_substr in k][0]
            val = (2.0 * pm[key]["FETCH_SIZE"]["avg_per_launch"] + pm[key]["WRITE_SIZE"]["avg_per_launch"]) * 1024.0
            return val, f"committed_profile:profiles/{summary}:{key[:60]}"
        except Exception:                                   # noqa: BLE001
            continue
    return None, None


def run_window(args, device, pix_dtype, window, shard=None, seed=None, state=None):
    """Build the window, warm up, capture, time exactly args.steps iterations; returns (wb, state, elapsed, graphed)."""
    import copy
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    cfg = copy.deepcopy(DEFAULT_CFG)
    cfg["photo_construction"]["nonmax_suppression_window"] = window
    if state is None:
        a2 = argparse.Namespace(**vars(args))
        if seed is not None:
            a2.seed = seed
        state = build_state(a2, device, pix_dtype)
    wb = WindowBA(state, cfg=cfg, pix_dtype=pix_dtype, window_full=True, shard=shard)
    for _ in range(args.warmup):
        wb.iterate()
    graphed = (not args.eager) and wb.capture()
    if shard is not None and getattr(shard, "world", 1) > 1 and not args.eager:
        # every rank must take the same path (graph replay or eager launches): if the capture failed on ANY rank, all run eager
        failed = shard.max_scalar(0.0 if graphed else 1.0, device)
        if failed > 0.0 and graphed:
            wb.graph = None
            wb.capture_error = "hipGraph capture failed on another rank"
            graphed = False
    return wb, state, graphed


def timed_steps(wb, steps, barrier=None):
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wb.step()
    torch.cuda.synchronize()
    if barrier is not None:
        barrier()
    return time.perf_counter() - t0


def block_kernel_roofline(wb, dtype_name, reps=5):
    """Dominant kernel (the pair-block kernel): duration from HIP events on the launch stream (events cannot sit inside a
    captured graph, so the same kernel on the same data is timed in eager iterations right after the timed region;
    profiles/ holds the rocprofv3 --kernel-trace --stats summary of the same command for cross-checking)."""
    wb.events = {}
    for _ in range(reps):
        wb.iterate()
    torch.cuda.synchronize()
    ev = wb.events.get("blocks", [])
    wb.events = None
    blk_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    pixel_pairs = wb.table.b * (0 if wb.idle else wb.n)          # wb.n = this rank's reference pixels per keyframe
    bytes_per = ALGO_SCALARS_PER_PIXEL_PAIR * (4 if dtype_name == "f32" else 8)
    achieved = pixel_pairs * bytes_per / (blk_ms * 1e-3) / 1e9 if blk_ms > 0 else 0.0
    kname = "ba_blocks_pair2_kernel<2,4>" if dtype_name == "f32" else "ba_blocks_pair2_f64_kernel<4,true,1,false>"
    traffic, src = committed_traffic("ba_blocks_pair2_kernel" if dtype_name == "f32" else "ba_blocks_pair2_f64")
    # the same launch priced on the matrix pipe: 320 v_mfma_*_16x16x4 (2048 flop each) per 64-pixel tile of a reference-keyframe
    # group (two pairs share the 10 depth x depth tiles); data-sheet dense peaks of guides/MI355X_MICROARCH.md, and -- float64 --
    # the bare-stream rate of scripts/micro/mfma_f64_rate.hip with every CU busy (profiles/r4_mfma_f64_rate.txt)
    ngrp = int(getattr(wb.table, "ngroups", 0) or 0)
    tiles = 0 if wb.idle else (wb.n + 63) // 64
    flops = ngrp * tiles * 320 * 2048.0
    peak_tf = 78.6 if dtype_name == "f64" else 157.3
    tf = flops / (blk_ms * 1e-3) / 1e12 if blk_ms > 0 else 0.0
    out = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src, "kernel_ms": blk_ms,
           "algorithmic_bytes_per_launch": pixel_pairs * bytes_per,
           "algorithmic_bytes_per_pixel_pair": bytes_per,
           "mfma_flop_per_launch": flops, "mfma_tflops": tf, "mfma_peak_tflops": peak_tf, "mfma_frac": tf / peak_tf}
    if dtype_name == "f64":
        # what a BARE back-to-back stream of the same instruction sustains with every CU busy (scripts/micro/mfma_f64_rate.hip, two
        # waves per SIMD; sclk logged at 2.39 GHz).  Not a ceiling of this kernel: interleaved with its vector work the stream runs
        # at the nominal 64 cycles per instruction and the bound is matrix + vector issue on one pipe
        try:
            for ln in open(os.path.join(ROOT, "profiles", "r4_mfma_f64_rate.txt")):
                if ln.startswith("f64 mfma 16x16x4 only") and "blocks= 512" in ln:
                    sus = float(ln.split("MFMA")[1].split("TFLOP/s")[0])
                    out["mfma_bare_stream_probe_tflops"] = sus
                    out["mfma_frac_of_bare_stream"] = tf / sus
                    out["mfma_probe_source"] = "committed_profile:profiles/r4_mfma_f64_rate.txt (scripts/micro/mfma_f64_rate.hip, two waves per SIMD)"
                    break
        except Exception:                                   # noqa: BLE001
            pass
    return out


def _lib_probe():
    from como_amd import _lib
    return _lib.lib().como_track_level_probe() == 1


def _tracking_level_us(device, H, W, steps=200):
    """microseconds per GN iteration of ONE persistent level launch at H x W (the pyramid levels of a 640x480 frame)."""
    import como_amd.odom.frontend.photo_tracking as pt
    from como_amd.utils import image_processing as ip
    tp = synth.make_tracking_pair(H=H, W=W, dtype=torch.float32, device=device, seed=3, levels=1)
    K = tp["intrinsics"]
    stack = ip.img_and_grads(tp["img_ref"])
    v, u = torch.meshgrid(torch.arange(float(H), device=device), torch.arange(float(W), device=device), indexing="ij")
    ray = torch.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)), -1).reshape(-1, 3)
    P = (tp["depth_ref"][0, 0].reshape(-1, 1) * ray)[None].contiguous()
    vals = tp["img_ref"].reshape(1, -1, 1).contiguous()
    dI = torch.stack((stack[0, 1].reshape(-1), stack[0, 2].reshape(-1)), -1)[None, :, None, :].contiguous()
    J = pt.precalc_jacobians(dI, P, vals, K)
    aff0 = torch.zeros((1, 2, 1), device=device)
    term = {"max_iter": steps, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}
    run = lambda: pt.photo_level_tracking(tp["Tji_init"], aff0, vals, P, J, tp["img_cur"], K, 0.1, term)
    run()
    best = None
    for _ in range(3):                                      # (fastest of three launches: a one-off ~70 ms stall of the first launch
        torch.cuda.synchronize()                            # after a change of image size was seen on two boxes)
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    rec = pt.photo_level_tracking.last_out.cpu()
    return best / steps * 1e6, (int(rec[105]) == steps and int(rec[104]) == 0)


def tracking_leg(device, steps=200):
    """Config 2: the tracking GN iteration on a synthetic 640x480 pair, level 0 (N = 307,200 reference pixels): the captured
    iteration graph of como_amd/odom/frontend/photo_tracking.py replayed `steps` times (float32 = the reference's tracking
    dtype).  Unit A: 53 B per pixel-iteration."""
    try:
        import como_amd.odom.frontend.photo_tracking as pt
        from como_amd.utils import image_processing as ip
        tp = synth.make_tracking_pair(H=480, W=640, dtype=torch.float32, device=device, seed=3, levels=1)
        K = tp["intrinsics"]
        stack = ip.img_and_grads(tp["img_ref"])
        v, u = torch.meshgrid(torch.arange(480., device=device), torch.arange(640., device=device), indexing="ij")
        ray = torch.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)), -1).reshape(-1, 3)
        P = (tp["depth_ref"][0, 0].reshape(-1, 1) * ray)[None].contiguous()
        vals = tp["img_ref"].reshape(1, -1, 1).contiguous()
        dI = torch.stack((stack[0, 1].reshape(-1), stack[0, 2].reshape(-1)), -1)[None, :, None, :].contiguous()
        J = pt.precalc_jacobians(dI, P, vals, K)
        aff0 = torch.zeros((1, 2, 1), device=device)
        term = {"max_iter": steps, "delta_norm": 0.0, "rel_tol": 0.0, "grad_norm": 0.0}     # exactly `steps` iterations
        run = lambda: pt.photo_level_tracking(tp["Tji_init"], aff0, vals, P, J, tp["img_cur"], K, 0.1, term)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Tn, _ = run()                                        # ONE launch: the persistent level kernel runs all iterations
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rec = pt.photo_level_tracking.last_out.cpu()
        graphed = False
        fused = int(rec[105]) == steps and int(rec[104]) == 0

        class _LG:                                           # (kept: the reporting below reads lg.T)
            T = Tn
        lg = _LG()
        us = el / steps * 1e6
        N = P.shape[1]
        algo = N * 53.0
        terr = (lg.T - tp["Tji_gt"]).abs().max().item()
        ach = algo / (us * 1e-6) / 1e9
        # HBM bytes per ITERATION from the committed counter pass of scripts/track_profile.py (one launch = `iterations_per_launch`
        # iterations of the same 640x480 level; FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE as is)
        tr_iter, tr_src = None, None
        try:
            tr_file = "r5_track_pmc_summary.json" if os.path.exists(os.path.join(ROOT, "profiles", "r5_track_pmc_summary.json")) else "r4_track_pmc_summary.json"
            pm = json.load(open(os.path.join(ROOT, "profiles", tr_file)))
            key = [k for k in pm if "track_level_kernel" in k][0]
            per_launch = (2.0 * pm[key]["FETCH_SIZE"]["avg_per_launch"] + pm[key]["WRITE_SIZE"]["avg_per_launch"]) * 1024.0
            tr_iter = per_launch / float(pm["_meta"]["track_iterations_per_launch"])
            tr_src = f"committed_profile:profiles/{tr_file}:" + key[:40]
        except Exception:                                   # noqa: BLE001
            pass
        # the coarser pyramid levels of the same frame size (one persistent launch each; a tracked frame runs all three)
        per_level = {"640x480": us}
        for (hh, ww) in ((240, 320), (120, 160), (60, 80)):       # (160x120 and 80x60 run XCD-local when the probe allows it)
            try:
                per_level[f"{ww}x{hh}"] = _tracking_level_us(device, hh, ww, steps)[0]
            except Exception as e:                          # noqa: BLE001
                per_level[f"{ww}x{hh}"] = repr(e)[:120]
        return {"workload": f"config 2: 2-frame 640x480 photometric tracking GN iteration, level 0, N={N} reference pixels, float32",
                "value": steps / el, "unit": "GN iters/s", "us_per_iter": us, "us_per_iter_by_level": per_level, "steps": steps,
                "hip_graph": bool(graphed),
                "persistent_level_kernel": bool(fused),
                "xcd_local_coarse_levels": bool(_lib_probe()),
                "pixels_per_s": N * steps / el, "max_pose_abs_err_vs_gt_end": terr,
                "roofline": {"bound": "hbm", "kernel": "track_level_kernel (per iteration)", "achieved": ach, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": tr_iter, "traffic_source": tr_src,
                             "algorithmic_bytes_per_launch": algo, "algorithmic_bytes_per_pixel": 53.0}}
    except Exception as e:                                  # noqa: BLE001
        return {"error": repr(e)[:300]}


def ate_leg(device):
    """BASELINE.json's "ATE vs ref": the HIP loop's trajectory against the one the REFERENCE's sequential loop produced on the
    same 72 frames (192x256, config/como.yml parameters; tests/golden/ate_sequence.npz, generated by running the reference)."""
    try:
        import numpy as np
        from como_amd.utils.ate import ate_rmse
        from scripts.ate_sequence import run_ate_sequence
        d = np.load(os.path.join(ROOT, "tests", "golden", "ate_sequence.npz"))
        G = {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fiub" else d[k]) for k in d.files}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kinds, poses, odo = run_ate_sequence(G, "float", str(device))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ref_kinds = [int(x) for x in G["kinds"]]
        tr = [k for k in range(len(ref_kinds)) if bool(G["tracked"][k]) and k in poses]
        est, ref, gt = [poses[k] for k in tr], [G["T_w_curr"][k] for k in tr], [G["poses_gt"][k] for k in tr]
        return {"workload": "72 rendered frames 192x256, config/como.yml parameters (9-keyframe sliding window, 24 one-way frames, m=64, "
                            "window 4, float32 tracking, float64 mapping system / float32 pixel kernels), seeded DepthCov weights",
                "value": ate_rmse(est, ref), "unit": "m (ATE-RMSE of the tracked positions vs the reference's trajectory, no alignment)",
                "ate_rmse_sim3_aligned": ate_rmse(est, ref, "sim3"), "frames": len(ref_kinds), "tracked": len(tr),
                "same_decisions": sum(int(a == b) for a, b in zip(kinds, ref_kinds)),
                "path_length_m": float(sum((G["poses_gt"][k + 1, :3, 3] - G["poses_gt"][k, :3, 3]).norm() for k in range(len(ref_kinds) - 1))),
                "ate_vs_ground_truth_sim3": {"hip": ate_rmse(est, gt, "sim3"), "reference": ate_rmse(ref, gt, "sim3")},
                "wall_s_including_setup": el}
    except Exception as e:                                  # noqa: BLE001
        return {"error": repr(e)[:300]}


def replicas_main(args, shard, device):
    """Config 5: one sequence per GPU (throughput mode).  Every rank runs its OWN rendered sequence through the whole odometry
    loop; no collective touches the data path (the barrier / max only bracket the timed region)."""
    pix = "double" if args.dtype == "f64" else "float"
    # warm-up run on a short sequence: library load, kernel code objects, the allocator's pools
    r = odometry_loop(device, seed=1 + shard.rank, pix=pix, warm=args.warmup, barrier=shard.barrier, timed_frames=args.steps)
    if "error" in r:
        raise SystemExit("replica sequence failed on rank %d: %s" % (shard.rank, r["error"]))
    elapsed = shard.max_scalar(r["elapsed_s"], device)
    per_rank = shard.gather_scalars(r["value"], device)
    dist_rec = cdist.dist_record(shard, device)
    if shard.rank == 0:
        out = {"metric": "frames/sec, one 640x480 sequence per GPU through the whole odometry loop (config 5, throughput mode)",
               "value": shard.world * args.steps / elapsed, "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": r["workload"] + "; one step = one frame (tracking, keyframe management, DepthCov network + "
                                      "sampler on every new keyframe, one window-BA GN iteration)",
                          "parallelism": "replicas: one sequence per GPU, no collective", "frames_per_rank": args.steps,
                          "keyframes_inserted_rank0": r["keyframes_inserted"], "one_way_inserted_rank0": r["one_way_inserted"],
                          "tracking_chain_fallbacks_rank0": r["tracking_chain_fallbacks"]},
               "per_rank_frames_per_s": per_rank, "dist": dist_rec, "git": git_sha()}
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    shard.barrier()
    import torch.distributed as tdist
    if tdist.is_initialized():
        tdist.destroy_process_group()


def self_launch(n, argv=None):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: re-run this command line as N ranks under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>` (the same
    launch the driver's N > 1 command uses) and return its exit status.  The ranks inherit stdout: rank 0 prints the ONE JSON
    line; the launcher itself only writes to stderr."""
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    with socket.socket() as s:                               # a port nobody listens on right now
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print("bench.py: no WORLD_SIZE in the environment -- starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--window", type=int, default=1, choices=[1, 2, 4, 8])
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"],
                    help="per-pixel path: f64 = the reference's mapping dtype (config/como.yml:28), f32 = mixed precision")
    ap.add_argument("--keyframes", type=int, default=8)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cpu-dense", action="store_true", help="cpu_baseline on the window-4 workload only (skips the two dense oracle iterations: ~20 s, ~30 GB of host memory)")
    ap.add_argument("--eager", action="store_true", help="do not capture the iteration into a hipGraph")
    ap.add_argument("--no-secondary", action="store_true", help="only the headline leg (no f32 / window-4 / tracking / odometry legs)")
    ap.add_argument("--replicas", action="store_true", help="config 5: one SEQUENCE per GPU through the whole odometry loop, no collective (weak scaling)")
    ap.add_argument("--force-shard", action="store_true", help="testing on a one-GPU box: run the multi-GPU code path (RCCL "
                    "collectives, sharded medians, fixed-point exchange) through a single-rank process group")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: nobody started the ranks -- start them here (one process per GPU under
        # torch.distributed.run, rendezvous on 127.0.0.1) and hand rank 0's single JSON line through
        raise SystemExit(self_launch(args.gpus))
    shard, device = cdist.init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    if shard.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={shard.world} (launch with --nproc-per-node {args.gpus}, or without "
                         f"torch.distributed.run: bench.py then starts its own ranks)")
    if args.replicas:
        return replicas_main(args, shard, device)
    pix_dtype = torch.float32 if args.dtype == "f32" else torch.float64
    sharded = shard.world > 1
    if args.force_shard and shard.world == 1:
        import torch.distributed as tdist
        if not tdist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            tdist.init_process_group("nccl", rank=0, world_size=1)
        shard = cdist.Shard(0, 1, force_collectives=True)
        sharded = True

    # ---- headline leg ----
    wb, state, graphed = run_window(args, device, pix_dtype, args.window, shard=(shard if sharded else None))
    if not graphed and not args.eager and shard.rank == 0 and not sharded:
        print("hipGraph capture failed:", getattr(wb, "capture_error", "?"), file=sys.stderr)
    elapsed = shard.max_scalar(timed_steps(wb, args.steps, barrier=shard.barrier), device)
    ms_step = elapsed / args.steps * 1e3
    roof = block_kernel_roofline(wb, args.dtype)
    import como_amd.odom.backend.linear_system as lin_sys
    info = int(lin_sys.solve_system.last_info)
    pose_err = (wb.kf_poses - state["poses_gt"]).abs().max().item()
    pose_err0 = (state["kf_poses"] - state["poses_gt"]).abs().max().item()
    npairs = wb.table.b
    # what the process group saw (every rank takes part): backend, devices, collectives per iteration, capture per rank, and where a
    # rank's eager iteration goes -- so that an N-GPU line shows by itself that RCCL ran on N distinct GPUs (como_amd/dist.py)
    dist_rec = cdist.dist_record(shard, device, wb, graphed)
    dist_rec["sharded_path"] = bool(sharded)

    legs, flat = {}, {}
    single = args.gpus == 1 and not args.no_secondary and args.window == 1 and args.keyframes == 8
    if single:
        # the other per-pixel dtype on the same dense window
        other = "f32" if args.dtype == "f64" else "f64"
        odt = torch.float64 if other == "f64" else torch.float32
        del wb
        torch.cuda.empty_cache()
        wbo, sto, go = run_window(args, device, odt, args.window)
        eo = timed_steps(wbo, args.steps)
        ro = block_kernel_roofline(wbo, other)
        name = "mixed_precision" if other == "f32" else "reference_dtype"
        legs[name] = {
            "workload": f"same window, per-pixel path in {other}" + (" (config/como.yml:28 mapping dtype double)" if other == "f64"
                                                                     else " (mixed precision: system / priors / solve / state stay float64)"),
            "dtype": other, "value": args.steps / eo, "unit": "GN iters/s", "ms_per_step": eo / args.steps * 1e3, "steps": args.steps,
            "hip_graph": bool(go), "roofline": ro,
            "max_pose_abs_err_vs_gt_end": (wbo.kf_poses - sto["poses_gt"]).abs().max().item()}
        flat[f"{name}_{other}_gn_iters_per_s"] = args.steps / eo
        flat[f"{name}_{other}_ms_per_step"] = eo / args.steps * 1e3
        del wbo, sto
        torch.cuda.empty_cache()
        # the reference's default sub-selection (config/como.yml: nonmax_suppression_window 4, n = 19,200 px / keyframe)
        sec = {}
        for nm, dt_ in (("f64", torch.float64), ("f32", torch.float32)):
            wb4, st4, g4 = run_window(args, device, dt_, 4)
            e4 = timed_steps(wb4, args.steps)
            sec[nm] = {"value": args.steps / e4, "ms_per_step": e4 / args.steps * 1e3, "hip_graph": bool(g4)}
            n4 = wb4.n
            del wb4, st4
            flat[f"window4_{nm}_gn_iters_per_s"] = sec[nm]["value"]
            flat[f"window4_{nm}_ms_per_step"] = sec[nm]["ms_per_step"]
        legs["secondary"] = {"workload": f"same window, nonmax_suppression_window=4 (n={n4} reference px/KF, the reference's default "
                                         f"= its operating point: config/como.yml:28,37)",
                             "unit": "GN iters/s", "value": sec["f64"]["value"], "ms_per_step": sec["f64"]["ms_per_step"],
                             "hip_graph": sec["f64"]["hip_graph"], "dtype": "f64", "f32": sec["f32"]}
        torch.cuda.empty_cache()
        legs["tracking"] = tracking_leg(device)
        # the sequence is run twice: the first pass meets every one-time cost of a process (hipGraph captures of the tracker frame
        # and the network, pinned staging blocks, the allocator growing to the window's final sizes: ~0.1 s in total, e.g. 15-40 ms
        # on the frame at which the window first fills) -- `value` is the steady-state rate of the second pass, the first pass's
        # rate is reported beside it
        # ... and the steady state is taken from THREE further passes (median): a single pass of 0.2 s is at the mercy of one host
        # hiccup (a 20-130 ms stall in one frame was seen in about one pass of twenty on the test boxes); all rates are in the line
        first = odometry_loop(device)
        steady = [odometry_loop(device) for _ in range(3)]
        ok = sorted((r for r in steady if "value" in r), key=lambda r: r["value"])
        legs["odometry_loop"] = ok[len(ok) // 2] if ok else steady[-1]
        if "value" in legs["odometry_loop"] and "value" in first:
            legs["odometry_loop"]["first_pass_frames_per_s"] = first["value"]
            legs["odometry_loop"]["steady_passes_frames_per_s"] = [r.get("value") for r in steady]
            legs["odometry_loop"]["protocol"] = ("median of three steady-state passes over the sequence in this process (steady_passes_frames_per_s) after a "
                                                 "first pass that meets the one-time captures / allocations (first_pass_frames_per_s)")
        if "value" in legs["odometry_loop"]:
            flat["odometry_loop_frames_per_s"] = legs["odometry_loop"]["value"]
        if "value" in legs["tracking"]:
            flat["tracking_us_per_iter"] = legs["tracking"]["us_per_iter"]
        legs["ate_vs_ref"] = ate_leg(device)
        if "value" in legs["ate_vs_ref"]:
            flat["ate_vs_ref_rmse_m"] = legs["ate_vs_ref"]["value"]

    if shard.rank == 0:
        mode = "dp%d (reference-pixel shards of every pair)" % args.gpus
        cfg = {"workload": f"{args.keyframes}-keyframe {args.width}x{args.height} window BA, {npairs} keyframe pairs, "
                           f"n={wb_n(state, args)} reference px/KF (window={args.window}), m=64, D={roof_dim(state, args)}; one step = "
                           f"full GN iteration (scaffold, dense ref, photometric system, priors, Cholesky solve, update)",
               "pixel_pairs_per_iter": npairs * wb_n(state, args), "system_dim": roof_dim(state, args), "pix_dtype": args.dtype,
               "system_dtype": "f64", "hip_graph": bool(graphed), "parallelism": mode}
        cfg.update(flat)
        out = {
            "metric": "GN iters/sec, 8-keyframe 640x480 photometric BA" if args.keyframes == 8 else
                      f"GN iters/sec, {args.keyframes}-keyframe {args.width}x{args.height} photometric BA",
            "value": args.steps / elapsed, "unit": "GN iters/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": cfg,
            "roofline": roof,
            "solution": {"cholesky_info": info, "max_pose_abs_err_vs_gt_start": pose_err0, "max_pose_abs_err_vs_gt_end": pose_err},
            "dist": dist_rec,
            "git": git_sha(),
        }
        out.update(legs)
        if not args.no_cpu and args.gpus == 1 and args.keyframes == 8:
            st_cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in state.items()}
            out["cpu_baseline"] = cpu_baseline(args, st_cpu)
        # RCCL writes its version banner through C stdio (flushed at exit): push it out first so that the JSON line is last
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                                   # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    shard.barrier()
    import torch.distributed as tdist
    if tdist.is_initialized():
        tdist.destroy_process_group()


def wb_n(state, args):
    return (args.height // args.window) * (args.width // args.window)


def roof_dim(state, args):
    return 8 * args.keyframes + 3 * int(state["P_m"].shape[0])


if __name__ == "__main__":
    main()
