#!/bin/bash
# Round 3, GPU call 4: float64 dense reference on the matrix cores, select tail without fallback launches, odometry loop host profile.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c4
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r3.py -x -q > $OUT/pytest_a.log 2>&1; echo "rc $?" >> $OUT/pytest_a.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python bench.py --dtype f64 --no-cpu --no-secondary --eager --steps 20 --warmup 3 > $OUT/prof_f64_run.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_f64_eager_kernel_stats.csv
timeout 150 python bench.py --dtype f64 --no-cpu --no-secondary --steps 20 --warmup 3 > $OUT/bench_f64.json 2> $OUT/bench_f64.err
timeout 150 python bench.py --dtype f64 --window 4 --no-cpu --no-secondary --steps 20 --warmup 3 > $OUT/bench_w4_f64.json 2> $OUT/bench_w4_f64.err
COMO_ODO_BREAKDOWN=1 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odo_breakdown.json 2> $OUT/odo_breakdown.err
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --cprofile-after 30 > $OUT/odo_cprof.json 2> $OUT/odo_cprof.err
cp gpurun_out/odo_cprofile.txt $OUT/odo_cprofile.txt
timeout 900 python -m pytest tests/test_gpu_r2.py -x -q > $OUT/pytest_b.log 2>&1; echo "rc $?" >> $OUT/pytest_b.log
python - <<'PY'
import json, glob, os, csv
for f in sorted(glob.glob("gpurun_out/r3c4/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), "it/s %.1f ms %.3f blk_ms %.4f frac %.3f poseerr %.6e info %s" % (
            d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            d["solution"]["max_pose_abs_err_vs_gt_end"], d["solution"]["cholesky_info"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", repr(e)[:100], open(f.replace(".json", ".err")).read()[-600:])
rows = list(csv.DictReader(open("gpurun_out/r3c4/bench_f64_eager_kernel_stats.csv")))
for r in rows[:14]:
    print("  %-64s calls %4d avg %8.1f us" % (r["Name"][:64], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
tail -3 $OUT/pytest_a.log; tail -3 $OUT/pytest_b.log
cut -c1-1200 $OUT/odo_breakdown.json
head -45 $OUT/odo_cprofile.txt | cut -c1-150
