"""The N > 1 protocol of como_amd/dist.py with world_size 2 on gloo (CPU): pixel-range shards, the distributed EXACT
median (per-digit histogram all-reduce) and the single all-reduce of the fixed-point per-pair sums.

The HIP kernels cannot run here; each rank emulates its kernel phases with the CPU oracle / numpy on ITS pixel range and
drives the real `Shard` collectives -- the result must equal the single-process result."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import load_golden

SHIFTS = {np.float32: (21, 10, 0), np.float64: (53, 42, 31, 20, 10, 0)}
BITS = {np.float32: (11, 11, 10), np.float64: (11, 11, 11, 11, 10, 10)}


def _keys(r):
    a = np.abs(r)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint64)


def _resolve(hists, npass, shifts):
    """Host mirror of csrc/select.cuh sel_resolve: (prefix, k_rem, nvalid) from finished digit histograms."""
    prefix, k = 0, 0
    nv = int(hists[0].sum())
    for p in range(npass):
        h = hists[p]
        if p == 0:
            k = (nv - 1) // 2 if nv > 0 else 0
        c = np.cumsum(h)
        b = int(np.searchsorted(c, k, side="right"))
        prefix |= b << shifts[p]
        k -= int(c[b - 1]) if b > 0 else 0
    return prefix, k, nv


def _distributed_median(shard, r_local, valid_local):
    dt = r_local.dtype.type
    shifts, bits = SHIFTS[dt], BITS[dt]
    keys = _keys(r_local)[valid_local].astype(np.uint64)
    hists = [None] * len(shifts)
    for p in range(len(shifts)):
        prefix, _, _ = _resolve(hists, p, shifts) if p else (0, 0, 0)
        sel = keys
        if p > 0:
            sh = shifts[p - 1]
            sel = keys[(keys >> np.uint64(sh)) == np.uint64(prefix >> sh)]
        digit = ((sel >> np.uint64(shifts[p])) & np.uint64((1 << bits[p]) - 1)).astype(np.int64)
        h = torch.from_numpy(np.bincount(digit, minlength=2048).astype(np.int32))
        shard.all_reduce_sum(h)                       # the collective of como_amd.dist
        hists[p] = h.numpy().astype(np.int64)
    prefix, _, nv = _resolve(hists, len(shifts), shifts)
    arr = np.array([prefix], dtype=np.uint32 if dt == np.float32 else np.uint64)
    return arr.view(dt)[0], nv


def _distributed_median_exchange(shard, r_local, valid_local, cap=512):
    """The float64 protocol of como_amd/odom/window_ba.py (csrc/select.hip como_select_cand_*): three all-reduced digit histograms
    (33 bits), then ONE all-gather of the keys that still match; every rank finishes digits 3..5 from the union."""
    shifts, bits = SHIFTS[np.float64], BITS[np.float64]
    keys = _keys(r_local)[valid_local].astype(np.uint64)
    hists = [None] * 6
    for p in range(3):
        prefix, _, _ = _resolve(hists, p, shifts) if p else (0, 0, 0)
        sel = keys
        if p > 0:
            sh = shifts[p - 1]
            sel = keys[(keys >> np.uint64(sh)) == np.uint64(prefix >> sh)]
        digit = ((sel >> np.uint64(shifts[p])) & np.uint64((1 << bits[p]) - 1)).astype(np.int64)
        h = torch.from_numpy(np.bincount(digit, minlength=2048).astype(np.int32))
        shard.all_reduce_sum(h)
        hists[p] = h.numpy().astype(np.int64)
    prefix, _, _ = _resolve(hists, 3, shifts)
    cand = keys[(keys >> np.uint64(shifts[2])) == np.uint64(prefix >> shifts[2])]
    assert cand.size <= cap
    rec = np.zeros(1 + cap, dtype=np.int64)
    rec[0] = cand.size
    rec[1:1 + cand.size] = cand.view(np.int64)
    loc = torch.from_numpy(rec)
    allr = torch.zeros((shard.world, 1 + cap), dtype=torch.int64)
    shard.all_gather(allr, loc)                       # the ONE exchange
    union = np.concatenate([allr[r, 1:1 + int(allr[r, 0])].numpy().view(np.uint64) for r in range(shard.world)])
    for p in range(3, 6):
        pfx, _, _ = _resolve(hists, p, shifts)
        sh = shifts[p - 1]
        sel = union[(union >> np.uint64(sh)) == np.uint64(pfx >> sh)]
        digit = ((sel >> np.uint64(shifts[p])) & np.uint64((1 << bits[p]) - 1)).astype(np.int64)
        hists[p] = np.bincount(digit, minlength=2048).astype(np.int64)
    prefix, _, nv = _resolve(hists, 6, shifts)
    return np.array([prefix], dtype=np.uint64).view(np.float64)[0], nv


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank)})
    from como_amd import dist as cdist
    from oracle import photo_ba
    shard, device = cdist.init_from_env(backend="gloo")
    try:
        out = {}
        # 1. distributed exact median on random data, float32 and float64, with ties
        for dt in (np.float32, np.float64):
            g = np.random.default_rng(5)
            n = 100_003
            r = (g.standard_normal(n) * np.exp(2 * g.standard_normal(n))).astype(dt)
            r[::7] = dt(0.125)
            valid = g.random(n) < 0.8
            b, e = shard.pixel_range(n)
            med, nv = _distributed_median(shard, r[b:e], valid[b:e])
            ref = torch.median(torch.from_numpy(np.abs(r[valid]))).item()
            out[f"median_{dt.__name__}"] = (float(med), ref, nv, int(valid.sum()))
            if dt is np.float64:                              # the 3 + 1 protocol gives the same key
                med2, nv2 = _distributed_median_exchange(shard, r[b:e], valid[b:e])
                out["median_float64_exchange"] = (float(med2), ref, nv2, int(valid.sum()))
        # 2. sharded normal equations of the golden window: each rank linearises its pixel range, ONE all-reduce
        G = load_golden("ba_window_f64.npz")
        rid, tid = G["kf_ref_ids"].long(), G["kf_target_ids"].long()
        n = G["Pwn"].shape[1]
        b, e = shard.pixel_range(n)
        sl = slice(b, e)
        r, valid, J = photo_ba.pair_rows(G["vals_n"][rid][:, sl], G["kf_aff_params"][rid], G["Pwn"][rid][:, sl], G["kf_poses"][tid],
                                         G["kf_aff_params"][tid], G["kf_img_and_grads"][tid], G["dPwn_dTwc"][rid][:, sl],
                                         G["dPwn_dzm"][rid][:, sl], G["intrinsics"][0])
        med, nv = _distributed_median(shard, r.numpy().reshape(-1), valid.numpy().reshape(-1))
        sigma = torch.tensor(1.4826, dtype=torch.float64) * float(med)
        Gm, gv, err = photo_ba.pair_blocks(r, valid, J, sigma)
        # the exchange of the product (csrc/ba.hip ba_reduce_assemble MODE 1 / 2): this rank's per-pair sums are split into
        # floor(v) and (v - floor(v)) 2^56, all-reduced as INTEGERS (exact, order-independent), and every rank expands the
        # same bits into the normal equations
        nb = Gm.shape[0]
        vec = torch.cat((Gm.reshape(nb, -1), gv.reshape(nb, -1)), dim=1).reshape(-1)
        vec = torch.cat((vec, err.reshape(1).double()))
        hi = torch.floor(vec)
        fix = torch.stack((hi.to(torch.int64), ((vec - hi) * 2.0 ** 56).to(torch.int64)), dim=1).contiguous()
        shard.all_reduce_sum(fix)
        tot = fix[:, 0].double() + fix[:, 1].double() * 2.0 ** -56
        Gm_t = tot[:-1].reshape(nb, -1)[:, :Gm[0].numel()].reshape(Gm.shape)
        gv_t = tot[:-1].reshape(nb, -1)[:, Gm[0].numel():].reshape(gv.shape)
        D = G["H_photo"].shape[0]
        H, gg = torch.zeros((D, D), dtype=torch.float64), torch.zeros(D, dtype=torch.float64)
        photo_ba.assemble(Gm_t, gv_t, G["dzm_dPwm"][rid], G["kf_inds"][rid], G["kf_inds"][tid], G["landmark_inds"][rid], H, gg)
        out["sigma"] = (float(sigma), float(G["sigma_r"]))
        out["H_rel"] = float((H - G["H_photo"]).abs().max() / G["H_photo"].abs().max())
        out["g_rel"] = float((gg - G["g_photo"]).abs().max() / G["g_photo"].abs().max())
        out["err"] = (float(tot[-1]), float(G["photo_err"]))
        out["H_bits"] = H.numpy().tobytes()
        out["range"] = (b, e, n)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_pixel_ranges_partition():
    from como_amd.dist import Shard
    for n in (64, 100, 19_200, 307_200, 12_345):
        for world in (1, 2, 3, 4, 8):
            cover = np.zeros(n, dtype=int)
            for r in range(world):
                b, e = Shard(r, world).pixel_range(n)
                assert 0 <= b <= e <= n                      # b == e: an idle rank (more ranks than 64-pixel tiles) owns nothing
                cover[b:e] += 1
            assert cover.min() == 1 and cover.max() == 1     # every pixel exactly once, whatever n and world


def test_two_rank_protocol_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        o = res[rank]
        for k in ("median_float32", "median_float64", "median_float64_exchange"):
            got, ref, nv, nvalid = o[k]
            assert got == ref and nv == nvalid               # exact, on every rank
        assert o["sigma"][0] == pytest.approx(o["sigma"][1], rel=1e-14)
        assert o["H_rel"] < 1e-12 and o["g_rel"] < 1e-12
        assert o["err"][0] == pytest.approx(o["err"][1], rel=1e-12)
    assert res[0]["range"][1] == res[1]["range"][0]          # contiguous shards
    assert res[0]["H_bits"] == res[1]["H_bits"]              # integer all-reduce: both ranks hold the same bits


def test_bench_self_launch_builds_the_drivers_command(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE must start its own ranks: the launcher re-runs the same arguments under
    torch.distributed.run with N processes on 127.0.0.1 (the driver's N > 1 command) -- checked here without a GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    assert bench.self_launch(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _record_worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)})
    from como_amd import dist as cdist
    shard, device = cdist.init_from_env(backend="gloo")
    try:
        h = torch.ones(8, dtype=torch.int32)
        for _ in range(3):
            shard.all_reduce_sum(h)
        shard.all_gather(torch.zeros((world, 4), dtype=torch.int32), torch.full((4,), rank, dtype=torch.int32))
        shard.all_reduce_sum(torch.ones(16, dtype=torch.int64))
        rec = cdist.dist_record(shard, device, graph_captured=False)
        q.put((rank, rec, shard.n_collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_dist_record_gathers_every_rank(world):
    """The record bench.py prints with a multi-GPU line (como_amd/dist.py dist_record), on CPU ranks: every rank answers the
    gather, the `Shard` wrappers count the data-path collectives (3 histogram all-reduces + the candidate all-gather + the sums =
    the five of a float64 iteration), every rank ends with the same record."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 300) + world
    procs = [ctx.Process(target=_record_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, rec, n = q.get(timeout=240)
        res[rank] = (rec, n)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        rec, n = res[rank]
        assert n == 5
        assert rec["backend"] == "gloo" and rec["world"] == world and rec["ranks_answered"] == list(range(world))
        assert [r["rank"] for r in rec["per_rank"]] == list(range(world))
        assert rec["graph_captured_all"] is False and rec["per_rank"] == res[0][0]["per_rank"]
