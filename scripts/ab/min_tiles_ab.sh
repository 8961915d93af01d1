#!/bin/bash
# GPU box: A/B of the per-CU-bounded lower limit on the tiles per pixel chunk of the two-pair block kernels (COMO_BA_MIN_TILES=8,
# como_amd/odom/backend/photo.py linearize): the window-4 bench legs and the pinned odometry loop, same box, two repetitions.
for rep in 1 2; do for mt in 0 8; do
COMO_BA_MIN_TILES=$mt python - <<'PY'
import os, sys, json, argparse, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
args = argparse.Namespace(keyframes=8, height=480, width=640, m=64, seed=0, eager=False, warmup=5, steps=20, window=4, dtype="f64")
out = {"min_tiles": os.environ.get("COMO_BA_MIN_TILES")}
for nm, dt in (("f64", torch.float64), ("f32", torch.float32)):
    try:
        wb, st, g = bench.run_window(args, dev, dt, 4)
        e = min(bench.timed_steps(wb, 20) for _ in range(3))
        out["window4_" + nm] = round(20 / e, 1)
        del wb, st
    except Exception as ex:
        out["window4_" + nm] = repr(ex)[:200]
r = bench.odometry_loop(dev)
out["loop_fps"] = round(r.get("value", -1), 1)
print(json.dumps(out), flush=True)
PY
done; done
