#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
COMO_ODO_BREAKDOWN=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_odo2 -- python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/ab/run.log 2>&1
F=$(find /tmp/p_odo2 -name "*kernel_trace.csv" | head -1)
if [ -n "$F" ]; then
python - "$F" <<'PY'
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
print("dispatches", len(rows), "columns", list(rows[0].keys())[:20])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r.get("Kernel_Name", "")[:90]
    g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[(name, g, wg)]
    a[0] += 1; a[1] += d
top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]
with open("gpurun_out/ab/by_grid.txt", "w") as f:
    for (name, g, wg), (c, t) in top:
        f.write(f"{t/1e3:8.2f} ms {c:6d} x {t/c:8.1f} us  grid {g:>9} wg {wg:>5}  {name}\n")
print(open("gpurun_out/ab/by_grid.txt").read()[:6000])
PY
fi
tail -1 gpurun_out/ab/run.log | cut -c1-200
