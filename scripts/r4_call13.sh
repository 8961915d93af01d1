mkdir -p gpurun_out/r4c13
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --cprofile-after 40 > /dev/null 2>&1
cp gpurun_out/odo_cprofile.txt gpurun_out/r4c13/
head -120 gpurun_out/odo_cprofile.txt | cut -c1-180
