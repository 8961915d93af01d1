mkdir -p gpurun_out/r4c16
timeout 600 python -m pytest tests/test_gpu_r4.py -q -x -k "weighted_normal" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py -q -x -k "corr or mapping or odometry or ate or track_and_init or sequential or distill" 2>&1 | tail -4
for i in 1 2; do COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-120; done
