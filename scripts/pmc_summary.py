"""Summarise rocprofv3 outputs into the small tracked files under profiles/.

    python scripts/pmc_summary.py <dir with *_counter_collection.csv / *_kernel_stats.csv> <out.json>

Per kernel: launches, average FETCH_SIZE / WRITE_SIZE (KB, as reported) from the --pmc passes.  The gfx950 correction of
guides/MI355X_MICROARCH.md (FETCH_SIZE reports half of the bytes of wide coalesced reads) is NOT applied here: consumers
(bench.py) apply it and say so.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").strip()
    return name


def main(src, out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    res = {}
    for k, cs in acc.items():
        res[k] = {c: {"avg_per_launch": v[0] / max(v[1], 1), "launches": v[1]} for c, v in cs.items()}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k in sorted(res):
        print(k[:70], {c: round(v["avg_per_launch"], 1) for c, v in res[k].items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
