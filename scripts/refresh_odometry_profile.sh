#!/bin/bash
# refresh only the odometry-loop part of the round-3 profiles (kernel stats under rocprofv3 + the unprofiled loop line)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r3; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_run.log 2>&1
F=$(find /tmp/p_odo -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/odometry_kernel_stats.csv
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop.json 2> $OUT/odometry_loop.err
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 > $OUT/odometry_loop300.json 2>> $OUT/odometry_loop.err
cut -c1-300 $OUT/odometry_loop.json; echo; cut -c1-300 $OUT/odometry_loop300.json; echo
head -12 $OUT/odometry_kernel_stats.csv | cut -c1-150
