mkdir -p gpurun_out/r4c9
timeout 900 python -m pytest tests/test_gpu_dist.py -q -x -k "config4 and 8" --tb=short > gpurun_out/r4c9/tests.txt 2>&1
grep -v "socket.cpp\|amdgpu.ids" gpurun_out/r4c9/tests.txt | tail -40
