#!/bin/bash
# Round 4, second collection (after the loop work of §4.16): only what that work changed -- the odometry loop (rocprofv3 stats,
# GPU timeline, unprofiled 100 / 300-frame lines), the small-system timings, config 5, the window-4 chain and the default bench
# line -> gpurun_out/profiles_r4b/.  The block / dense-reference / solver / network kernels are unchanged since
# collect_profiles_r4.sh ran; their files stay.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r4b
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_run.log 2>&1
cp $(find /tmp/p_odo -name "*kernel_stats.csv" | head -1) $OUT/odometry_kernel_stats.csv
python scripts/odometry_timeline.py /tmp/p_odo $OUT/odometry_timeline.txt > /dev/null 2>&1
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop.json 2> $OUT/odometry_loop.err
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 > $OUT/odometry_loop300.json 2>> $OUT/odometry_loop.err
for F in 1 0; do echo "COMO_CHOL_SMALL_FAST=$F"; COMO_CHOL_SMALL_FAST=$F timeout 120 python scripts/chol_small_time.py; done > $OUT/chol_small_time.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_w4 -- python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype f64 --window 4 > $OUT/bench_w4_stats_run.log 2>&1
cp $(find /tmp/p_w4 -name "*kernel_stats.csv" | head -1) $OUT/bench_w4_f64_eager_kernel_stats.csv
COMO_SINGLE_DEVICE=1 COMO_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --replicas --gpus 2 --steps 60 --warmup 5 > $OUT/bench_replicas2_one_gpu.json 2> $OUT/bench_replicas2_one_gpu.err
timeout 300 python bench.py --replicas --steps 60 --warmup 5 > $OUT/bench_replicas1.json 2> $OUT/bench_replicas1.err
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
head -12 $OUT/odometry_timeline.txt
cut -c1-330 $OUT/odometry_loop.json; echo; cut -c1-200 $OUT/odometry_loop300.json; echo
for f in $OUT/bench_replicas2_one_gpu.json $OUT/bench_replicas1.json; do tail -1 $f | cut -c1-330; echo; done
tail -1 $OUT/bench_line.json | cut -c1-1200
