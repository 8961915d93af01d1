#!/bin/bash
# GPU box (through gpurun): the round-5 odometry timeline on the reference-pinned 640x480 sequence.
#   bash scripts/collect_profiles_r5.sh odo     -> gpurun_out/r5p/odometry_timeline.txt, odometry_kernel_stats.csv, odometry_loop.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5p; mkdir -p $OUT
case "$1" in
odo)
  COMO_ODO_BREAKDOWN=1 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop_parts.json 2> $OUT/odo_parts.err
  rm -rf /tmp/p_odo; COMO_ODO_BREAKDOWN=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop.json 2> $OUT/odo.err
  python scripts/odometry_timeline.py /tmp/p_odo $OUT/odometry_timeline.txt > /dev/null 2>&1
  F=$(find /tmp/p_odo -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -60 "$F" > $OUT/odometry_kernel_stats.csv
  ;;
esac
