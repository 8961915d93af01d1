#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3c14
COMO_ODO_BREAKDOWN=0 timeout 400 python scripts/gpu_odometry_bench.py --frames 100 --census-after 40 > gpurun_out/r3c14/odo.json 2> gpurun_out/r3c14/odo.err
tail -2 gpurun_out/r3c14/odo.err; cut -c1-400 gpurun_out/r3c14/odo.json; cp gpurun_out/odo_census.txt gpurun_out/r3c14/ 2>/dev/null; head -5 gpurun_out/r3c14/odo_census.txt
