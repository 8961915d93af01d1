import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "normalize_kernel" in r["Kernel_Name"]]
a, b = idx[-1], len(rows)
tot = {}
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    name = r["Kernel_Name"].replace("void ", "").replace("como::", "")[:34]
    tot[name] = tot.get(name, 0) + d
    if d > 25:
        print("%8.1f us grid %6s x %3s wg %4s  %s" % (d, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], name))
print(sorted(((round(v, 1), k) for k, v in tot.items()), reverse=True)[:8])
print("span %.1f us" % ((int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1000))
