mkdir -p gpurun_out/r4c12
timeout 600 python -m pytest tests/test_gpu_r3.py -q -x -k "small_spd" 2>&1 | tail -5
for f in 1 0; do echo "COMO_CHOL_SMALL_FAST=$f"; COMO_CHOL_SMALL_FAST=$f timeout 120 python scripts/chol_small_time.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c12/chol_small_time.txt
timeout 900 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py -q -x -k "prep_predictor or distill or corr or mapping or odometry or ate or own_conditioning or sampler" 2>&1 | tail -5
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-200
