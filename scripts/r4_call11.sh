mkdir -p gpurun_out/r4c11
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-300
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --census-after 40 > /dev/null 2>&1
cp gpurun_out/odo_census.txt gpurun_out/r4c11/odo_census.txt
head -90 gpurun_out/r4c11/odo_census.txt
timeout 600 python -m pytest tests/test_gpu_hotpath.py -q -x -k "odometry or tracker or mapping or ate" 2>&1 | tail -3
