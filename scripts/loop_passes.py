import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
for i in range(3):
    r = bench.odometry_loop(dev)
    print(round(r["value"], 1), {k: round(v["mean_ms"], 2) for k, v in r["frame_ms_by_request"].items()}, r.get("vs_reference_loop", {}).get("same_decisions"))
import como_amd.odom.window_ba as _w
print("median ahead:", _w.AHEAD_STATS)
