mkdir -p gpurun_out/r4_suite
timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > gpurun_out/r4_suite/pytest.txt 2>&1
grep -v "socket.cpp\|amdgpu.ids" gpurun_out/r4_suite/pytest.txt | tail -30
