"""The data-parallel window BA on real kernels: two processes (one-process-per-GPU layout) share the single test GPU and
exchange through `gloo` (RCCL refuses two ranks on one device; the collectives are the same `Shard` calls the `nccl`
run issues).  Each rank linearises its pixel range of every pair with the HIP kernels; after the histogram and
normal-equation all-reduces both must hold the single-process result."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0"})
    import torch.distributed as dist
    from como_amd import dist as cdist
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA
    shard, device = cdist.init_from_env(backend="gloo")
    try:
        def predictor(cov, cm):
            Kinv, L, Kt = prep_predictor(cov.double(), cm.double(), 1.0)
            return Kinv, L, Kt.float()
        out = {}
        for use_shard in (False, True):
            st = synth.make_window(B=4, H=96, W=128, m=16, dtype=torch.float64, device=device, seed=3, predictor=predictor)
            wb = WindowBA(st, pix_dtype=torch.float32, window_full=True, shard=(shard if use_shard else None))
            wb.iterate()
            wb.iterate()
            torch.cuda.synchronize()
            out["sharded" if use_shard else "single"] = (wb.kf_poses.cpu().numpy().copy(), wb.H.cpu().numpy().copy(), float(wb.err))
        q.put((rank, out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_sharded_window_ba_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        ps, Hs, es = res[rank]["single"]
        pd, Hd, ed = res[rank]["sharded"]
        # float32 per-pixel partial sums regroup with the shard boundaries: agreement at float32 round-off level
        assert np.abs(Hd - Hs).max() / np.abs(Hs).max() < 2e-6
        assert abs(ed - es) / abs(es) < 3e-4            # cost: float32 running sum per workgroup, regrouped by the shards
        assert np.abs(pd - ps).max() < 1e-6
    # both ranks hold the SAME BITS: the shards' per-pair sums are all-reduced as fixed-point integers (exact), the expansion,
    # the priors (exact integer atomics) and the solve are replicated deterministically
    assert np.array_equal(res[0]["sharded"][1], res[1]["sharded"][1])          # H
    assert np.array_equal(res[0]["sharded"][0], res[1]["sharded"][0])          # poses after two iterations
    # ... and the single-process result is bit-identical on both ranks too (no order-dependent accumulation anywhere)
    assert np.array_equal(res[0]["single"][1], res[1]["single"][1]) and np.array_equal(res[0]["single"][0], res[1]["single"][0])


def _config4_worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0"})
    import copy
    import torch.distributed as dist
    from como_amd import dist as cdist
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA, DEFAULT_CFG
    from tests.conftest import load_golden
    shard, device = cdist.init_from_env(backend="gloo")
    try:
        G = load_golden("ba_window32_f64.npz")

        def predictor(cov, cm):                      # the reference's K_mm^-1 (cond ~ 1e8), as tests/test_gpu_r2.py
            Kinv, L, Kt = prep_predictor(cov.double(), cm.double(), 1.0, K_mm_inv=G["K_mm_inv"])
            if L is None:
                L = torch.linalg.cholesky(torch.linalg.inv(Kinv))
            return Kinv, L, Kt
        st = synth.make_window(B=int(G["B"]), H=int(G["H"]), W=int(G["W"]), m=int(G["m"]), dtype=torch.float64, device=device,
                               seed=int(G["seed"]), predictor=predictor, aff_noise=float(G["aff_noise"]))
        cfg = copy.deepcopy(DEFAULT_CFG)
        cfg["photo_construction"]["nonmax_suppression_window"] = int(G["window"])
        wb = WindowBA(st, cfg=cfg, pix_dtype=torch.float64, window_full=True, shard=shard)
        # count the data-path collectives of every iteration (the `Shard` calls the RCCL run issues)
        calls = {"n": 0}
        for name in ("all_reduce_sum", "all_reduce_max", "all_gather"):
            def wrap(f):
                def g(*a, **k):
                    calls["n"] += 1
                    return f(*a, **k)
                return g
            setattr(shard, name, wrap(getattr(shard, name)))
        out, ncoll = [], []
        for it in range(3):
            calls["n"] = 0
            wb.iterate()
            torch.cuda.synchronize()
            ncoll.append(calls["n"])
            if it < 2:
                out.append((wb.kf_poses.cpu().numpy().copy(), wb.P_m.cpu().numpy().copy(), int(wb.sigma[1]), wb.H.cpu().numpy().copy()))
        q.put((rank, out, wb.table.b, wb.dim, 0 if wb.idle else wb.n, wb.n_total, ncoll))    # (an idle rank owns no pixel)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_config4_window_vs_reference(world):
    """Config 4's shape (a 32-keyframe window, 62 pairs, D = 8 B + 3 L) in the one-process-per-GPU mode: `world` ranks (8 = the
    target node, emulated on the one GPU), each with its pixel range of every pair, exchange the median histograms / candidates
    and the fixed-point per-pair sums; every rank must end with the reference's Mapping.iterate result
    (tests/golden/ba_window32_f64.npz), all ranks with identical bits, in at most FIVE collectives per iteration."""
    from tests.conftest import load_golden, report
    G = load_golden("ba_window32_f64.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 300) + world
    procs = [ctx.Process(target=_config4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    shares = 0
    for _ in range(world):
        rank, out, b, dim, n, n_total, ncoll = q.get(timeout=900)
        res[rank] = out
        assert b == 62 and dim == G["it0_g_full"].shape[0] and 0 <= n < n_total         # a share of the pixels (world 8: idle ranks own none)
        shares += n
        # float64 keys: 3 histogram all-reduces + ONE candidate all-gather for both exact medians + the per-pair sums; the first
        # iteration also all-reduces the row-norm bound of the band median once
        assert ncoll[0] <= 6 and ncoll[1] <= 5 and ncoll[2] <= 5, ncoll
    assert shares == n_total
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    worst = {"pose": 0.0, "P": 0.0}
    for it in range(2):
        poses, P, nvalid, H = res[0][it]
        assert nvalid == int(G[f"it{it}_pair_nvalid"].sum())                             # the global valid count: exact
        worst["pose"] = max(worst["pose"], float(np.abs(poses - G[f"it{it}_kf_poses_new"].numpy()).max()))
        worst["P"] = max(worst["P"], float(np.abs(P - G[f"it{it}_P_new"].numpy()).max()))
        for r in range(1, world):                                                        # every rank: the same bits
            assert np.array_equal(res[r][it][0], poses) and np.array_equal(res[r][it][3], H)
    report("sharded_config4", world=world, **worst)
    assert worst["pose"] < 1e-8 and worst["P"] < 1e-6


def _nccl_worker(port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    import torch.distributed as dist
    from como_amd import synth
    from como_amd.dist import Shard
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1)            # backend "nccl" IS RCCL on ROCm
    try:
        def predictor(cov, cm):
            Kinv, L, Kt = prep_predictor(cov.double(), cm.double(), 1.0)
            return Kinv, L, Kt.float()
        out = {}
        def predictor64(cov, cm):
            return prep_predictor(cov.double(), cm.double(), 1.0)
        for name, sh, pix in (("single", None, torch.float32), ("rccl", Shard(0, 1, force_collectives=True), torch.float32),
                              ("single64", None, torch.float64), ("rccl64", Shard(0, 1, force_collectives=True), torch.float64)):
            st = synth.make_window(B=4, H=96, W=128, m=16, dtype=torch.float64, device=device, seed=3,
                                   predictor=predictor if pix == torch.float32 else predictor64)
            wb = WindowBA(st, pix_dtype=pix, window_full=True, shard=sh)
            wb.iterate()
            wb.iterate()
            graphed = wb.capture(warmup=1)
            for _ in range(2):
                wb.step()
            torch.cuda.synchronize()
            out[name] = (wb.kf_poses.cpu().numpy().copy(), wb.H.cpu().numpy().copy(), bool(graphed),
                         getattr(wb, "capture_error", "")[-400:])
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_single_rank_group():
    """The sharded iteration through a real `nccl` (= RCCL) process group of ONE rank (the test box has one GPU; RCCL refuses
    two ranks on a device): int32 histogram and int64 fixed-point all-reduces on the GPU, issued eagerly and inside a
    captured hipGraph.  With one rank the shard is the whole window, so the result must equal the unsharded one BITWISE."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(29900 + (os.getpid() % 90), q))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    from tests.conftest import report
    report("rccl_single_rank", graph_single=out["single"][2], graph_rccl=out["rccl"][2], capture_error=out["rccl"][3])
    dpose = float(np.abs(out["single"][0] - out["rccl"][0]).max())
    dH = float(np.abs(out["single"][1] - out["rccl"][1]).max())
    assert np.array_equal(out["single"][0], out["rccl"][0]) and np.array_equal(out["single"][1], out["rccl"][1]), (dpose, dH)
    assert out["single"][2]                                             # the unsharded iteration always captures
    # float64 per-pixel path: three int32 all-reduces + the candidate all-gather (all_gather_into_tensor on RCCL), eager and captured
    assert np.array_equal(out["single64"][0], out["rccl64"][0]) and np.array_equal(out["single64"][1], out["rccl64"][1])
    report("rccl_single_rank_f64", graph_single=out["single64"][2], graph_rccl=out["rccl64"][2], capture_error=out["rccl64"][3])


def test_bench_two_ranks_share_one_gpu():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), on the one-GPU test
    rig: both ranks on device 0, gloo instead of RCCL (COMO_SINGLE_DEVICE / COMO_DIST_BACKEND).  Checks the whole multi-rank
    flow -- sharded window, collectives, eager fallback where the backend cannot be captured, max-over-ranks timing, ONE JSON
    line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, COMO_SINGLE_DEVICE="1", COMO_DIST_BACKEND="gloo")
    port = 29700 + (os.getpid() % 200)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert d["metric"] and d["roofline"]["kernel_ms"] > 0 and d["solution"]["cholesky_info"] == 0
    assert d["solution"]["max_pose_abs_err_vs_gt_end"] < d["solution"]["max_pose_abs_err_vs_gt_start"]
    # the self-verifying record of the process group (como_amd/dist.py dist_record)
    rec = d["dist"]
    assert rec["backend"] == "gloo" and rec["world"] == 2 and rec["ranks_answered"] == [0, 1] and rec["sharded_path"] is True
    assert rec["distinct_devices"] == 1                       # (the one-GPU rig: an N-GPU node must report N here)
    assert rec["collectives_per_iteration"] == [5]            # float64: 3 histogram all-reduces + 1 candidate all-gather + 1 sums
    assert rec["graph_captured_all"] is False                 # gloo stages through the host: eager on every rank
    for r in rec["per_rank"]:
        t = r["eager_us_per_iteration"]
        assert t["total"] > 0 and t["collectives"] > 0 and t["replicated_tail"] > 0
        assert abs(t["total"] - t["sharded_and_setup"] - t["collectives"] - t["replicated_tail"]) < 1e-6 * t["total"]


def test_bench_self_launch_without_world_size():
    """`python bench.py --gpus 2` with NO WORLD_SIZE / RANK in the environment (the shape of the driver's N = 1 command): bench.py
    must start its two ranks itself (self_launch -> torch.distributed.run on 127.0.0.1) and still print exactly ONE JSON line.
    One-GPU test rig as above."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(COMO_SINGLE_DEVICE="1", COMO_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=root,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "starting 2 ranks" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["solution"]["cholesky_info"] == 0
    assert d["dist"]["world"] == 2 and d["dist"]["ranks_answered"] == [0, 1]


def _dist_record_worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0"})
    import torch.distributed as dist
    from como_amd import dist as cdist
    from como_amd import synth
    from como_amd.depth_cov.core.covariance import prep_predictor
    from como_amd.odom.window_ba import WindowBA
    shard, device = cdist.init_from_env(backend="gloo")
    try:
        def predictor(cov, cm):
            return prep_predictor(cov.double(), cm.double(), 1.0)
        st = synth.make_window(B=4, H=96, W=128, m=16, dtype=torch.float64, device=device, seed=3, predictor=predictor)
        wb = WindowBA(st, pix_dtype=torch.float64, window_full=True, shard=shard)
        wb.iterate()
        rec = cdist.dist_record(shard, device, wb, graph_captured=False, eager_iters=2)
        q.put((rank, rec))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_dist_record_world8():
    """The record the bench line carries for a multi-GPU run, gathered over an EIGHT-rank group (the target node's world, emulated
    on the one GPU with gloo): every rank answers, every rank reports the same five collectives per float64 iteration and a
    consistent split of its eager iteration, every rank ends with the same record."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 300)
    procs = [ctx.Process(target=_dist_record_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(8))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in range(8):
        rec = res[rank]
        assert rec["backend"] == "gloo" and rec["world"] == 8 and rec["ranks_answered"] == list(range(8))
        assert rec["collectives_per_iteration"] == [5] and rec["distinct_devices"] == 1 and rec["graph_captured_all"] is False
        assert [r["rank"] for r in rec["per_rank"]] == list(range(8))
        for r in rec["per_rank"]:
            t = r["eager_us_per_iteration"]
            assert t["total"] > 0 and t["replicated_tail"] > 0 and t["collectives"] > 0
    assert all(res[r]["per_rank"] == res[0]["per_rank"] for r in range(8))


def test_bench_replicas_one_sequence_per_rank():
    """BASELINE config 5 (one SEQUENCE per GPU, throughput mode) as `bench.py --replicas` runs it: every rank drives its own
    rendered 640x480 sequence through the whole odometry loop (tracking, keyframe management, DepthCov network + sampler on
    every new keyframe, one window-BA iteration per frame), no data-path collective.  One-GPU test rig: both ranks on device 0
    (their kernels interleave), gloo for the bracketing barrier / max."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, COMO_SINGLE_DEVICE="1", COMO_DIST_BACKEND="gloo")
    port = 29400 + (os.getpid() % 200)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--replicas", "--gpus", "2",
                        "--steps", "30", "--warmup", "2"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert len(d["per_rank_frames_per_s"]) == 2 and min(d["per_rank_frames_per_s"]) > 0
    assert d["value"] > 0 and abs(d["value"] - 2 * 30 / (d["ms_per_step"] * 30 / 1e3)) < 1e-6 * d["value"]
    assert d["config"]["frames_per_rank"] == 30
    assert d["dist"]["world"] == 2 and d["dist"]["ranks_answered"] == [0, 1] and d["dist"]["backend"] == "gloo"


def test_replica_sequences_share_one_gpu():
    """Config 5's correctness: two replica PROCESSES on the one GPU, started 0 s and 1.5 s after their contexts are up so that
    their phases (two-frame initialisation, keyframe insertions, persistent tracking launches) collide differently, each running
    the 72-frame sequence of tests/golden/ate_sequence.npz through the whole loop.  Each trajectory must match the REFERENCE's own
    (same request on every frame, ATE-RMSE < 2e-5 m) -- in particular when another process's kernels keep the persistent tracking
    kernel's workgroups from being co-resident (its barrier then gives up and the frame falls back to the per-iteration chain)."""
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.mkdtemp()
    procs = []
    for rank, delay in ((0, 0.0), (1, 1.5)):
        out = os.path.join(tmp, f"replica{rank}.json")
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(root, "scripts", "replica_ate_check.py"), "--rank", str(rank),
                                             "--delay", str(delay), "--out", out], cwd=root, stdout=subprocess.PIPE,
                                            stderr=subprocess.PIPE, text=True)))
    res = []
    for out, p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-2000:]
        res.append(json.load(open(out)))
    for r in res:
        assert r["tracked"] >= 60 and r["same_decisions"] == r["frames"] and r["kf_timestamps_equal"], r
        assert r["ate_rmse"] < 2e-5 and r["worst_pose_abs"] < 5e-5, r
    # the two runs really overlapped in time
    assert res[1]["delay"] < res[0]["seconds"], res
