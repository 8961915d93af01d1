"""GPU timeline of the odometry loop from a rocprofv3 --kernel-trace csv: busy time (union of the kernel intervals), the gaps
between kernels by length class, and the kernels that precede the long gaps -- over the frames after the two-frame
initialisation (from the first track_level_kernel on).   python scripts/odometry_timeline.py <dir> [out.txt] [compact.csv]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as c:
        t0 = rows[0][0]
        for s, e, n in rows:
            c.write("%d,%d,%s\n" % (s - t0, e - t0, n[:70].replace(",", ";")))
first = next(i for i, r in enumerate(rows) if "track_level" in r[2])
post = rows[first:]
nfr = sum(1 for r in post if "track_level" in r[2]) / 3.0
span = (max(r[1] for r in post) - post[0][0]) / 1e3
dur = sum(e - s for s, e, _ in post) / 1e3
busy, cur_s, cur_e = 0, post[0][0], post[0][1]
gaps = []
for s, e, n in post[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e3, n, last))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last = n
busy = (busy + cur_e - cur_s) / 1e3
print("post-init: %d dispatches over %.1f tracked frames: span %.1f us/frame, busy %.1f us/frame (%.0f %%), sum of durations %.1f us/frame, "
      "%.0f dispatches/frame" % (len(post), nfr, span / nfr, busy / nfr, 100 * busy / span, dur / nfr, len(post) / nfr), file=out)
def shares(rs, label):
    tot = sum(e - s for s, e, _ in rs) / 1e6
    como = sum(e - s for s, e, n in rs if "como::" in n) / 1e6
    blas = sum(e - s for s, e, n in rs if n.startswith("Cijk")) / 1e6
    cp = sum(e - s for s, e, n in rs if "copyBuffer" in n or "fillBuffer" in n) / 1e6
    print("%s: %d dispatches, %.1f ms of kernel time: como:: %.1f ms (%.0f %%), library GEMMs %.1f ms, copy / fill %.1f ms, other torch "
          "kernels %.1f ms" % (label, len(rs), tot, como, 100 * como / tot, blas, cp, tot - como - blas - cp), file=out)


shares(rows, "whole run")
shares(post, "after the initialisation")
# frames by kind: a frame = the dispatches from one level-0 tracking launch to the next
tl = [i for i, r in enumerate(rows) if "track_level" in r[2]][::3]
kinds = {"plain tracked frame": [], "one-way frame + window rebuild": [], "keyframe insertion": []}
for a, b in zip(tl[:-1], tl[1:]):
    prev, fb = rows[a][0], 0
    for s, e, _ in rows[a:b]:
        if e > prev:
            fb += e - max(s, prev)
            prev = e
    # (by content, not by dispatch count: a keyframe insertion runs the covariance network, a one-way insertion the float64
    # image stack of Mapping.add_one_way_frame -- round 6: frame_stack_kernel; before: img_grads_kernel<double>)
    names = [n for _, _, n in rows[a:b]]
    k = ("keyframe insertion" if any("conv3_" in n or "conv_mfma" in n for n in names) else
         ("one-way frame + window rebuild" if any("img_grads_kernel<double>" in n or "frame_stack_kernel" in n for n in names)
          else "plain tracked frame"))
    kinds[k].append((fb / 1e3, b - a))
for k, v in kinds.items():
    if v:
        print("%-32s %3d frames: GPU busy %7.0f us, %5.0f dispatches per frame" % (k, len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)), file=out)
print("", file=out)
cls = [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 300), (300, 1000), (1000, 1e9)]
for lo, hi in cls:
    g = [x for x in gaps if lo <= x[0] < hi]
    print("gaps %5g..%-5g us: %6d  total %8.1f us/frame" % (lo, hi, len(g), sum(x[0] for x in g) / nfr), file=out)
print("\nkernels FOLLOWING the gaps >= 20 us (who the GPU waited for), by total gap:", file=out)
agg = collections.defaultdict(lambda: [0, 0.0])
for g, n, l in gaps:
    if g >= 20:
        a = agg[n[:80]]
        a[0] += 1
        a[1] += g
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us/frame %5d x %7.1f us  %s" % (t / nfr, c, t / c, n), file=out)
print("\nkernels PRECEDING the gaps >= 20 us:", file=out)
agg = collections.defaultdict(lambda: [0, 0.0])
for g, n, l in gaps:
    if g >= 20:
        a = agg[l[:80]]
        a[0] += 1
        a[1] += g
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%8.1f us/frame %5d x %7.1f us  %s" % (t / nfr, c, t / c, n), file=out)
print("\nkernel time after the initialisation, us/frame:", file=out)
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in post:
    a = agg[n[:100]]
    a[0] += 1
    a[1] += (e - s) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%8.1f us/frame %7.2f /frame x %7.1f us  %s" % (t / nfr, c / nfr, t / c, n), file=out)
