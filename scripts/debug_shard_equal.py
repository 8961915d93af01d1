"""Development aid: is the sharded chain (single-rank nccl group) bit-identical to the unsharded one, eager and replayed from a
hipGraph?  Toggles: band median, side stream.  python scripts/debug_shard_equal.py"""
import os, sys
os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29917", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from como_amd import synth
from como_amd.dist import Shard
from como_amd.depth_cov.core.covariance import prep_predictor
from como_amd.odom.backend import dense_ref
from como_amd.odom.window_ba import WindowBA
device = torch.device("cuda:0")
torch.cuda.set_device(device)
dist.init_process_group("nccl", rank=0, world_size=1)


def predictor(cov, cm):
    Kinv, L, Kt = prep_predictor(cov.double(), cm.double(), 1.0)
    return Kinv, L, Kt.float()


def run(sh, graph, overlap):
    st = synth.make_window(B=4, H=96, W=128, m=16, dtype=torch.float64, device=device, seed=3, predictor=predictor)
    wb = WindowBA(st, pix_dtype=torch.float32, window_full=True, shard=sh)
    wb.overlap_priors = overlap
    wb.iterate(); wb.iterate()
    g = False
    if graph:
        g = wb.capture(warmup=1)
    else:
        wb.iterate()                                       # (the capture's warm-up iteration)
    for _ in range(2):
        wb.step()
    torch.cuda.synchronize()
    return wb.kf_poses.clone(), wb.H.clone(), wb.median_depths.clone(), g


for band in (True, False):
    for overlap in (True, False):
        dense_ref.BAND_MEDIAN = band
        ref = run(None, False, overlap)
        for name, sh in (("single", None), ("rccl", Shard(0, 1, force_collectives=True))):
            for graph in (False, True):
                r = run(sh, graph, overlap)
                print(f"band={band} overlap={overlap} {name} graph={graph} captured={r[3]} poses_eq={torch.equal(r[0], ref[0])} "
                      f"H_eq={torch.equal(r[1], ref[1])} med_eq={torch.equal(r[2], ref[2])} dpose={(r[0] - ref[0]).abs().max().item():.3e} "
                      f"dmed={(r[2] - ref[2]).abs().max().item():.3e}")
dist.destroy_process_group()
