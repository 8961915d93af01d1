#!/bin/bash
# ride-along back-substitution: parity tests of the solver + timing A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_gpu_hotpath.py -m gpu -q -k "cholesky" > gpurun_out/ab/pytest_chol.log 2>&1; tail -5 gpurun_out/ab/pytest_chol.log
timeout 300 python scripts/chol_time.py 760 200 1240 2>&1 | tee gpurun_out/ab/chol_ride.log
COMO_CHOL_RIDE_MAX=0 timeout 300 python scripts/chol_time.py 760 200 1240 2>&1 | tee gpurun_out/ab/chol_noride.log
COMO_CHOL_RIDE_MAX=0 timeout 600 python -m pytest tests/test_gpu_hotpath.py -m gpu -q -k "cholesky" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu --no-secondary --steps 20 --warmup 3 > gpurun_out/ab/bench_f64.json 2> gpurun_out/ab/bench_f64.err; python - <<'PY'
import json
for l in open("gpurun_out/ab/bench_f64.json"):
    if l.startswith("{"):
        d=json.loads(l); print("f64", d["value"], d["ms_per_step"])
PY
timeout 600 python bench.py --no-cpu --no-secondary --steps 20 --warmup 3 --window 4 > gpurun_out/ab/bench_w4.json 2> gpurun_out/ab/bench_w4.err; python - <<'PY'
import json
for l in open("gpurun_out/ab/bench_w4.json"):
    if l.startswith("{"):
        d=json.loads(l); print("w4 f64", d["value"], d["ms_per_step"])
PY
