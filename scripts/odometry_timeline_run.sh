#!/bin/bash
# kernel trace of the pinned odometry loop -> gpurun_out/odo_tl/{timeline.txt, compact.csv (start, end, kernel per dispatch)}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/odo_tl; mkdir -p $OUT; rm -rf /tmp/p_odo_tl
COMO_ODO_BREAKDOWN=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_odo_tl -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/run.log 2>&1
python scripts/odometry_timeline.py /tmp/p_odo_tl $OUT/timeline.txt $OUT/compact.csv > /dev/null 2>&1
gzip -f $OUT/compact.csv
head -8 $OUT/timeline.txt
