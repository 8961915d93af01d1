#!/bin/bash
# Round 3, GPU call 5: fused chain for the filling window (mean-log-depth scale prior in win_priors), select tail test, odometry loop.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c5
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_r2.py tests/test_gpu_hotpath.py -x -q -k "select or mapping or Mapping or sequential or odometry or ate or sfm or window" > $OUT/pytest_a.log 2>&1; echo "rc $?" >> $OUT/pytest_a.log
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --cprofile-after 30 > $OUT/odo_cprof.json 2> $OUT/odo_cprof.err
cp gpurun_out/odo_cprofile.txt $OUT/odo_cprofile.txt
COMO_ODO_BREAKDOWN=1 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odo_breakdown.json 2> $OUT/odo_breakdown.err
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest_a.log; tail -4 $OUT/pytest.log
cut -c1-1500 $OUT/odo_cprof.json; echo; cut -c1-1500 $OUT/odo_breakdown.json; echo
head -50 $OUT/odo_cprofile.txt | cut -c1-150
