"""Print the kernel timeline (duration, gap to the previous kernel) of the last GN iteration found in a rocprofv3
--kernel-trace csv:  python scripts/kernel_timeline.py <dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "win_update" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
prev = None
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1000 if prev else 0.0
    prev = e
    tot += (e - s) / 1000
    print("%8.1f us  gap %6.1f  %-58s grid %s" % ((e - s) / 1000, gap, r["Kernel_Name"][:58], r.get("Grid_Size_X", "?")))
print("sum of kernel durations %.1f us, span %.1f us" % (tot, (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1000))
