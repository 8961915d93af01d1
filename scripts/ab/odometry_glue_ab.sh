#!/bin/bash
# A/B of three host-glue variants of the odometry loop (round 3).  The COMO_AB_* switches existed only for this measurement: the
# variants lost and were removed, so this script documents the protocol rather than re-running it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
run() { COMO_ODO_BREAKDOWN=0 timeout 200 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | python -c "import sys,json; print('$1', round(json.loads(sys.stdin.read())['loop_fps_after_init'],1))"; }
for rep in 1 2; do
COMO_AB_PYR=1 COMO_AB_DEC=1 COMO_AB_SEQ=1 run all_on
COMO_AB_PYR=0 COMO_AB_DEC=0 COMO_AB_SEQ=0 run all_off
COMO_AB_PYR=1 COMO_AB_DEC=0 COMO_AB_SEQ=0 run pyr_only
COMO_AB_PYR=0 COMO_AB_DEC=1 COMO_AB_SEQ=0 run dec_only
COMO_AB_PYR=0 COMO_AB_DEC=0 COMO_AB_SEQ=1 run seq_only
done 2>&1 | tee gpurun_out/ab/ab.log
