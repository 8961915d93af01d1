cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4d2
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dist.py::test_sharded_config4_window_vs_reference 2>&1 | tail -8 > gpurun_out/r4d2/tests.txt
cat gpurun_out/r4d2/tests.txt
for i in 1 2; do COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-420; done | tee gpurun_out/r4d2/loop100.txt
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 2>/dev/null | tail -1 | cut -c1-300 | tee gpurun_out/r4d2/loop300.txt
COMO_ODO_BREAKDOWN=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_odo2 -- python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/r4d2/run.log 2>&1
python scripts/odometry_timeline.py /tmp/p_odo2 gpurun_out/r4d2/timeline.txt gpurun_out/r4d2/compact.csv
head -3 gpurun_out/r4d2/timeline.txt
