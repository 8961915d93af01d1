mkdir -p gpurun_out/r4c17
timeout 600 python -m pytest tests/test_gpu_hotpath.py -q -x -k "nn_ or depthcov or network or run_model" 2>&1 | tail -3
timeout 300 python scripts/nn_time.py --layers 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c17/nn_time.txt
