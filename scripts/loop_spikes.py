"""Per-pass frame-time spikes of the sequential loop: N passes over the pinned 640x480 sequence in one process, per pass the
rate after the initialisation and the slowest frames (ms, frame index, request).   python scripts/loop_spikes.py [passes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from como_amd import synth                                                     # noqa: E402
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule              # noqa: E402
from como_amd.odom.sequential import ComoSeq                                   # noqa: E402
from scripts.ate_sequence import SEQ640, loop_cfgs, render_frames              # noqa: E402

dev = "cuda:0"
G = dict(SEQ640, seed=1, nframes=100)
K, T, rgbs_cpu = render_frames(G)
rgbs = [r.to(dev) for r in rgbs_cpu]
model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
for p in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if len(sys.argv) > 2:
        import gc
        gc.collect()
    odo = ComoSeq(loop_cfgs(G, "float", dev, graph_network=True), K.clone(), (G["H"], G["W"]), model)
    ts, kinds = [], []
    for k in range(100):
        ts.append(time.perf_counter())
        kinds.append(odo.iter(1.0 + k, rgbs[k]))
    torch.cuda.synchronize()
    ts.append(time.perf_counter())
    k0 = 3
    fr = sorted(((1e3 * (ts[k + 1] - ts[k]), k, str(kinds[k])) for k in range(k0, 100)), reverse=True)
    ow = sorted((t for t, k, kd in fr if kd == "one-way"), reverse=True)
    print(f"pass {p}: {(100 - k0) / (ts[100] - ts[k0]):6.1f} frames/s; slowest", [(round(t, 2), k, kd) for t, k, kd in fr[:4]],
          "slowest one-way", [round(t, 2) for t in ow[:3]], "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2), flush=True)
