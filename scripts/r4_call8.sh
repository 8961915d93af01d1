mkdir -p gpurun_out/r4c8
timeout 1500 python -m pytest tests/test_gpu_r4.py tests/test_gpu_dist.py -q -x 2>&1 | tail -15 | tee gpurun_out/r4c8/tests.txt
