cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4d3
timeout 600 python -m pytest tests/test_gpu_r4.py tests/test_gpu_r3.py tests/test_gpu_hotpath.py tests/test_gpu_r2.py -m gpu -x -q -k "inherited or band_median or window_iterate or mapping_state or sequential_odometry or two_frame_init or ate_vs or tracker_glue or one_way or fullsize_metric" 2>&1 | tail -6 | tee gpurun_out/r4d3/tests.txt
for P in 1 0 1 0; do echo "COMO_SIDE_PRIORITY=$P"; COMO_SIDE_PRIORITY=$P COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-130; done | tee gpurun_out/r4d3/loop100.txt
echo "COMO_BA_REUSE_WS=0"; COMO_BA_REUSE_WS=0 COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-130 | tee -a gpurun_out/r4d3/loop100.txt
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 2>/dev/null | tail -1 | cut -c1-300 | tee gpurun_out/r4d3/loop300.txt
COMO_ODO_BREAKDOWN=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_odo2 -- python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/r4d3/run.log 2>&1
python scripts/odometry_timeline.py /tmp/p_odo2 gpurun_out/r4d3/timeline.txt gpurun_out/r4d3/compact.csv
head -3 gpurun_out/r4d3/timeline.txt
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | cut -c1-900
