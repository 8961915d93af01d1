#!/bin/bash
# Round 3, GPU call 1: wave -> SIMD placement probe, A/B of the float64 block-kernel variants on the dense window (compact dense
# reference), float32 check, rocprofv3 kernel stats of the default, then the GPU test suite.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c1
mkdir -p $OUT
scripts/micro/bin/simd_place > $OUT/simd_place.txt 2>&1
for V in 0 6 7 8 5 4 3 1; do
  COMO_BA_VARIANT=$V timeout 150 python bench.py --dtype f64 --no-cpu --no-secondary --steps 20 --warmup 3 > $OUT/bench_f64_v$V.json 2> $OUT/bench_f64_v$V.err
done
for V in 0 1; do
  COMO_BA_VARIANT=$V timeout 150 python bench.py --dtype f32 --no-cpu --no-secondary --steps 20 --warmup 3 > $OUT/bench_f32_v$V.json 2> $OUT/bench_f32_v$V.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python bench.py --dtype f64 --no-cpu --no-secondary --eager --steps 20 --warmup 3 > $OUT/prof_f64_run.log 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $OUT/bench_f64_eager_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python bench.py --dtype f32 --no-cpu --no-secondary --eager --steps 20 --warmup 3 > $OUT/prof_f32_run.log 2>&1
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $OUT/bench_f32_eager_kernel_stats.csv
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r3c1/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), "it/s %.1f ms %.3f blk_ms %.4f frac %.3f poseerr %.6e info %s" % (
            d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            d["solution"]["max_pose_abs_err_vs_gt_end"], d["solution"]["cholesky_info"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", repr(e)[:100], open(f.replace(".json", ".err")).read()[-300:])
PY
cat $OUT/simd_place.txt
head -12 $OUT/bench_f64_eager_kernel_stats.csv | cut -c1-150
tail -5 $OUT/pytest.log
