mkdir -p gpurun_out/r4c14
timeout 900 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py -q -x -k "tracker or mapping or odometry or ate or sfm_init or corr or track_and_init or sequential or state_machine" 2>&1 | tail -4
for i in 1 2; do COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-120; done
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 2>/dev/null | tail -1 | cut -c1-120
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --pix double 2>/dev/null | tail -1 | cut -c1-120
