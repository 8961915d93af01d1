#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py tests/test_gpu_r3.py -m gpu -q -x -k "track or state_machine or sequential_odometry or ate_ or glue" > gpurun_out/ab/pytest.log 2>&1; tail -15 gpurun_out/ab/pytest.log | cut -c1-300
run() { COMO_ODO_BREAKDOWN=0 timeout 200 python scripts/gpu_odometry_bench.py --frames 100 2>gpurun_out/ab/err_$1.log | python -c "import sys,json; print('$1', round(json.loads(sys.stdin.read())['loop_fps_after_init'],1))"; }
for rep in 1 2; do
COMO_TRACK_FRAME_GRAPH=1 run graph
COMO_TRACK_FRAME_GRAPH=0 run eager
done 2>&1 | tee gpurun_out/ab/ab.log
tail -3 gpurun_out/ab/err_graph.log | cut -c1-300
