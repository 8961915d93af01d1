"""A/B of the coarsest tracking level (80x60 of a 640x480 frame): the ONE-workgroup level kernel (csrc/track.hip
track_level_one_kernel, 1024 / 512 threads) against the multi-workgroup (XCD-local) form -- microseconds per iteration from one
launch of `steps` iterations, the best and the median of five launches each.  python scripts/track_one_ab.py [steps]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import _tracking_level_us  # noqa: E402
from como_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
L = _lib.lib()
res = {}
for name, nt in (("multi", 0), ("one1024", 1024), ("one768", 768), ("one512", 512), ("multi_again", 0), ("one768_again", 768)):
    L.como_track_level_set_one(nt)
    us = [_tracking_level_us(dev, 60, 80, steps)[0] for _ in range(5)]
    res[name] = {"best": min(us), "median": statistics.median(us)}
L.como_track_level_set_one(1024)
res["160x120_unchanged"] = _tracking_level_us(dev, 120, 160, steps)[0]
print(json.dumps(res))
