#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3c17
timeout 900 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py -m gpu -q -k "greedy or sampler or state_machine or sequential_odometry or ate_" > gpurun_out/r3c17/pytest.log 2>&1; tail -6 gpurun_out/r3c17/pytest.log
COMO_ODO_BREAKDOWN=0 timeout 400 python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/r3c17/odo.json 2> gpurun_out/r3c17/odo.err
cut -c1-330 gpurun_out/r3c17/odo.json
