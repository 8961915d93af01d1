"""GPU box: a noise-robust reading of the sequential loop for A/B switches (environment variables are read at import: one process per
setting).  N passes over the pinned 640x480 sequence in this process; the first is dropped (one-time captures); reported per frame
kind: the MEDIAN frame time over all remaining passes, and the loop time those medians add up to (97 frames) -- a keyframe frame
that is 0.5 ms late moves the plain frames/s of one pass by 0.4 %, the medians do not see it.
    python scripts/loop_ab.py [passes]"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
rates, med = [], {}
for i in range(n):
    r = bench.odometry_loop(dev)
    if "error" in r:
        print(r)
        sys.exit(1)
    if i == 0:
        continue
    rates.append(r["value"])
    for k, v in r["frame_ms_by_request"].items():
        med.setdefault(k, []).append((v["median_ms"], v["frames"]))
tot = 0.0
out = {}
for k, v in med.items():
    m = sorted(x[0] for x in v)[len(v) // 2]
    out[k] = round(m, 3)
    tot += m * v[0][1]
nfr = sum(v[0][1] for v in med.values())
print("passes (frames/s):", [round(x, 1) for x in rates], "| median frame ms by kind:", out,
      "| loop from the medians: %.1f frames/s" % (1e3 * nfr / tot), "| decisions", r.get("vs_reference_loop", {}).get("same_decisions"))
