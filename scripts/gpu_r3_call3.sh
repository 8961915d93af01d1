#!/bin/bash
# Round 3, GPU call 3: band median (window 4), float64 block-kernel ring depths, float64 MFMA rate probe.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c3
mkdir -p $OUT
scripts/micro/bin/mfma_f64_rate > $OUT/mfma_f64_rate.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_r3.py -x -q -k "band" > $OUT/pytest_band.log 2>&1; echo "rc $?" >> $OUT/pytest_band.log
for V in 0 9; do
  COMO_BA_VARIANT=$V timeout 150 python bench.py --dtype f64 --no-cpu --no-secondary --steps 20 --warmup 3 > $OUT/bench_f64_v$V.json 2> $OUT/bench_f64_v$V.err
done
for BM in 1 0; do
  for DT in f64 f32; do
    COMO_BAND_MEDIAN=$BM timeout 150 python bench.py --dtype $DT --window 4 --no-cpu --no-secondary --steps 20 --warmup 3 > $OUT/bench_w4_${DT}_band$BM.json 2> $OUT/bench_w4_${DT}_band$BM.err
  done
done
timeout 900 python -m pytest tests/test_gpu_r2.py -x -q -k "fullsize" > $OUT/pytest_fullsize.log 2>&1; echo "rc $?" >> $OUT/pytest_fullsize.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r3c3/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), "it/s %.1f ms %.3f blk_ms %.4f frac %.3f poseerr %.6e info %s" % (
            d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            d["solution"]["max_pose_abs_err_vs_gt_end"], d["solution"]["cholesky_info"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", repr(e)[:100], open(f.replace(".json", ".err")).read()[-600:])
PY
cat $OUT/mfma_f64_rate.txt
tail -4 $OUT/pytest_band.log; grep -h "band_median" $OUT/pytest_band.log | cut -c1-400; tail -4 $OUT/pytest_fullsize.log
