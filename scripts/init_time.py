"""Wall time of the two-frame initialiser's alignments (TwoFrameSfm.align_frame = the coarse-to-fine Gauss-Newton loop of
two_frame_sfm.py) on the first frames of the pinned 640x480 sequence, synchronising timers, after a warm-up pass over the same
frames.  COMO_SFM_PER_ITERATION=1 selects round 4's loop (A/B).     python scripts/init_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from como_amd import synth  # noqa: E402
from como_amd.depth_cov.core.DepthCovModule import DepthCovModule  # noqa: E402
from como_amd.odom.sequential import ComoSeq  # noqa: E402
import como_amd.odom.frontend.two_frame_sfm as tf  # noqa: E402
from scripts.ate_sequence import SEQ640, loop_cfgs, render_frames  # noqa: E402

dev = "cuda:0"
G = dict(SEQ640, seed=1, nframes=4)
K, T, rgbs = render_frames(G)
rgbs = [r.to(dev) for r in rgbs]
model = DepthCovModule({k: v.to(dev) for k, v in synth.depthcov_state_dict(0).items()})
for rep in range(2):                                       # (first pass: graph capture of the network, allocations)
    odo = ComoSeq(loop_cfgs(G, "float", dev, graph_network=True), K.clone(), (G["H"], G["W"]), model)
    sfm = odo.mapping.two_frame_sfm
    inner = sfm.align_frame
    times, iters = [], []
    inner_level = tf.two_frame_sfm

    def level(*a, **k):
        r = inner_level(*a, **k)
        iters.append(tf.two_frame_sfm.last_iters)
        return r

    def timed(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = inner(*a, **k)
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
        return r
    sfm.align_frame = timed
    tf.two_frame_sfm = level
    tf.two_frame_sfm.last_iters = 0
    try:
        for k in range(4):
            odo.iter(1.0 + k, rgbs[k])
    finally:
        tf.two_frame_sfm = inner_level
print("per-iteration loop (round 4)" if tf._LEGACY_LOOP else "round 5 loop", "-- align_frame ms per attempt:", [round(t, 2) for t in times],
      "GN iterations per level:", iters)
