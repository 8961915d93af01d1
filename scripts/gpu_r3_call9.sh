#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3c12
timeout 600 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py -m gpu -q -k "cholesky" > gpurun_out/r3c12/pytest_chol.log 2>&1; tail -15 gpurun_out/r3c12/pytest_chol.log
timeout 300 python scripts/chol_time.py 760 200 1240 2680 2>&1 | tee gpurun_out/r3c12/chol_ride.log
COMO_CHOL_RIDE_MAX=0 timeout 300 python scripts/chol_time.py 760 2>&1 | tee gpurun_out/r3c12/chol_noride.log
