mkdir -p gpurun_out/r4c15
timeout 1200 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py tests/test_gpu_r3.py -q -x -k "window or mapping or odometry or ate or rgb or band or reinit or compact or iterate or fused" 2>&1 | tail -4
for i in 1 2; do COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-120; done
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 2>/dev/null | tail -1 | cut -c1-120
