cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4d5
timeout 700 python -m pytest tests/test_gpu_r4.py tests/test_gpu_r3.py tests/test_gpu_hotpath.py tests/test_gpu_r2.py -m gpu -x -q -k "reproject or inherited or greedy or track_and_init or mapping_state or sequential_odometry or two_frame_init or ate_vs or ate_rgb or tracker_glue or tracking_state or como_backends or distill or replica" 2>&1 | tail -6 | tee gpurun_out/r4d5/tests.txt
for i in 1 2 3; do COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-330; done | tee gpurun_out/r4d5/loop100.txt
echo "COMO_GREEDY_FUSED=0"; COMO_GREEDY_FUSED=0 COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-330 | tee -a gpurun_out/r4d5/loop100.txt
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 2>/dev/null | tail -1 | cut -c1-300 | tee gpurun_out/r4d5/loop300.txt
COMO_ODO_BREAKDOWN=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_odo2 -- python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/r4d5/run.log 2>&1
python scripts/odometry_timeline.py /tmp/p_odo2 gpurun_out/r4d5/timeline.txt gpurun_out/r4d5/compact.csv
head -3 gpurun_out/r4d5/timeline.txt
