"""Development aid (GPU box): the pinned 640x480 sequence with the DepthCov network's output perturbed by `eps` relative noise:
where do the loop's decisions leave the reference's, and what does the trajectory do there?   python scripts/perturb_diag.py eps [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import como_amd.depth_cov.core.DepthCovModule as dcm
from como_amd import synth
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule
from como_amd.odom.sequential import ComoSeq
from scripts.ate_sequence import KIND_CODE, loop_cfgs, render_frames

eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ate_sequence_640.npz"))
G = {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fiub" else d[k]) for k in d.files}
g = torch.Generator(device=dev).manual_seed(seed)
calls = [0]


def hook(cov):
    calls[0] += 1
    if eps == 0.0:
        return cov
    if eps < 0.0:                                          # (a copy only: does anything depend on the identity of the network's output?)
        return cov.clone()
    nz = torch.randn((cov.shape[0], 3) + tuple(cov.shape[2:]), generator=g, device=cov.device, dtype=cov.dtype)
    return cov * (1.0 + eps * nz[:, [0, 1, 1, 2]])        # (E = [x, s, s, z] stays symmetric, as every output of the network is)


dcm.OUTPUT_HOOK = hook
K, T, rgbs = render_frames(G)
model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
odo = ComoSeq(loop_cfgs(G, "float", dev), K.clone(), (int(G["H"]), int(G["W"])), model)
ref_kinds = [int(x) for x in G["kinds"]]
kinds = []
for k in range(len(rgbs)):
    nb = len(odo.est_poses)
    try:
        kinds.append(KIND_CODE[odo.iter(1.0 + k, rgbs[k].to(dev))])
    except Exception as e:  # noqa: BLE001
        import traceback
        print(f"frame {k}: EXCEPTION {repr(e)[:200]}")
        traceback.print_exc()
        break
    err = float("nan")
    if len(odo.est_poses) > nb and bool(G["tracked"][k]):
        err = (odo.est_poses[-1].detach().cpu().double().reshape(4, 4) - G["T_w_curr"][k]).abs().max().item()
    flag = "" if kinds[-1] == ref_kinds[k] else f"   <-- reference: {ref_kinds[k]}"
    print(f"frame {k}: kind {kinds[-1]} |pose - ref|max {err:.3e} keyframes {len(odo.mapping.kf_timestamps)} landmarks "
          f"{int(odo.mapping.P_m.shape[0]) if odo.mapping.is_init else 0}{flag}", flush=True)
print("network calls", calls[0], "eps", eps)
