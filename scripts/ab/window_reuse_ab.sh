#!/bin/bash
# A/B: WindowBA(prev=...) takes over the keyframe-set-only topology on one-way insertions (COMO_BA_REUSE=0 rebuilds everything)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py tests/test_gpu_r3.py -m gpu -q -x -k "state_machine or sequential_odometry or ate_ or one_way" > gpurun_out/ab/pytest_reuse.log 2>&1; tail -6 gpurun_out/ab/pytest_reuse.log | cut -c1-300
run() { COMO_ODO_BREAKDOWN=0 timeout 200 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | python -c "import sys,json; print('$1', round(json.loads(sys.stdin.read())['loop_fps_after_init'],1))"; }
for rep in 1 2 3; do
COMO_BA_REUSE=1 run reuse
COMO_BA_REUSE=0 run rebuild
done 2>&1 | tee gpurun_out/ab/reuse_ab.log
