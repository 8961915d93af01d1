#!/bin/bash
# Round 3, GPU call 6: fused tracker glue (trackref.hip), whole GPU suite, odometry loop, default bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c6
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_r3.py -x -q -k "tracker_glue or band" > $OUT/pytest_a.log 2>&1; echo "rc $?" >> $OUT/pytest_a.log
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --cprofile-after 30 > $OUT/odo_cprof.json 2> $OUT/odo_cprof.err
cp gpurun_out/odo_cprofile.txt $OUT/odo_cprofile.txt
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -4 $OUT/pytest_a.log; tail -12 $OUT/pytest.log | cut -c1-300
cut -c1-900 $OUT/odo_cprof.json; echo
head -40 $OUT/odo_cprofile.txt | cut -c1-150
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c6/bench_default.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
print({k: v for k, v in d["config"].items() if k != "workload"})
print("odometry", d["odometry_loop"]); print("ate", d["ate_vs_ref"].get("value"), d["ate_vs_ref"].get("same_decisions"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["window4"]["value"])
PY
