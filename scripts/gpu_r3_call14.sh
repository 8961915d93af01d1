#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3c18
timeout 900 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py tests/test_gpu_r3.py -m gpu -q -k "track or state_machine or sequential_odometry or ate_ or glue" > gpurun_out/r3c18/pytest.log 2>&1; tail -6 gpurun_out/r3c18/pytest.log
COMO_ODO_BREAKDOWN=0 timeout 400 python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/r3c18/odo.json 2> gpurun_out/r3c18/odo.err
cut -c1-330 gpurun_out/r3c18/odo.json
COMO_ODO_BREAKDOWN=0 timeout 400 python scripts/gpu_odometry_bench.py --frames 300 > gpurun_out/r3c18/odo300.json 2>> gpurun_out/r3c18/odo.err
cut -c1-330 gpurun_out/r3c18/odo300.json
