cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4d1
timeout 300 python -m pytest tests/test_gpu_r4.py -k weighted_normal -x -q 2>&1 | tail -3
COMO_ODO_BREAKDOWN=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_odo2 -- python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/r4d1/run.log 2>&1
python scripts/odometry_timeline.py /tmp/p_odo2 gpurun_out/r4d1/timeline.txt gpurun_out/r4d1/compact.csv
head -12 gpurun_out/r4d1/timeline.txt
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>/dev/null | tail -1 | cut -c1-300
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --census-after 40 2>/dev/null | tail -1 | cut -c1-100
cp gpurun_out/odo_census.txt gpurun_out/r4d1/census.txt
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 --cprofile-after 40 2>/dev/null | tail -1 | cut -c1-100
cp gpurun_out/odo_cprofile.txt gpurun_out/r4d1/cprofile.txt
