"""GPU box: where WindowBA.__init__ / _prepare_topology / _prepare_fused (the window rebuild of a keyframe frame: ~0.9 ms of Python with
the device idle for ~0.3 ms of it) spend their host time -- cProfile of the second pass over the pinned sequence, callees of those
functions by own time.   python scripts/kf_init_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402

dev = torch.device("cuda:0")
bench.odometry_loop(dev)
pr = cProfile.Profile()
pr.enable()
r = bench.odometry_loop(dev)
pr.disable()
print(round(r["value"], 1), "frames/s under cProfile")
st = pstats.Stats(pr)
st.sort_stats("tottime")
for fn in ("__init__", "_prepare_topology", "_finish_topology", "_prepare_fused", "_host_tables", "_load_frames", "update_kf_reference",
           "add_keyframe", "track_and_init"):
    print("=" * 20, fn)
    st.print_callees(r"(window_ba|Mapping|Tracking|corr)\.py:\d+\(%s\)" % fn)
