#!/bin/bash
# Round 4 profile collection on the GPU box (through gpurun): rocprofv3 kernel stats + HBM / MFMA counters of the bench command (both
# per-pixel dtypes, eager launches so that every kernel of a GN iteration is its own record), of the window-4 leg, of the tracking
# level kernel and of the odometry loop -> gpurun_out/profiles_r4/ (the small summaries are copied into profiles/ afterwards).
# Counters are collected in their own passes (--kernel-trace + --pmc only).
set -u
R=r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
for DT in f64 f32; do
  CMD="python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype $DT"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats_$DT -- $CMD > $OUT/bench_${DT}_stats_run.log 2>&1
  cp $(find /tmp/p_stats_$DT -name "*kernel_stats.csv" | head -1) $OUT/bench_${DT}_eager_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch_$DT -- $CMD > $OUT/bench_${DT}_fetch_run.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write_$DT -- $CMD > $OUT/bench_${DT}_write_run.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma_f64 -- python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype f64 > $OUT/bench_f64_mfma_run.log 2>&1
mkdir -p /tmp/p_all /tmp/p_m
i=0; for f in $(find /tmp/p_fetch_f32 /tmp/p_write_f32 /tmp/p_fetch_f64 /tmp/p_write_f64 -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_all/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_all $OUT/bench_pmc_summary.json > $OUT/bench_pmc_summary.txt 2>&1
i=0; for f in $(find /tmp/p_mfma_f64 -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_m/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_m $OUT/bench_mfma_summary.json > $OUT/bench_mfma_summary.txt 2>&1
# the window-4 leg (the reference's operating point), float64
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_w4 -- python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype f64 --window 4 > $OUT/bench_w4_stats_run.log 2>&1
cp $(find /tmp/p_w4 -name "*kernel_stats.csv" | head -1) $OUT/bench_w4_f64_eager_kernel_stats.csv
# tracking level kernel: one launch = 200 iterations
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trk -- python scripts/track_leg.py 200 > $OUT/track_stats_run.log 2>&1
cp $(find /tmp/p_trk -name "*kernel_stats.csv" | head -1) $OUT/track_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_trk_f -- python scripts/track_leg.py 200 > $OUT/track_fetch_run.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_trk_w -- python scripts/track_leg.py 200 > $OUT/track_write_run.log 2>&1
mkdir -p /tmp/p_trk_all; i=0; for f in $(find /tmp/p_trk_f /tmp/p_trk_w -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_trk_all/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_trk_all $OUT/track_pmc_summary.json > $OUT/track_pmc_summary.txt 2>&1
python - <<'PY'
import json
p = "gpurun_out/profiles_r4/track_pmc_summary.json"
d = json.load(open(p))
# two launches per run (warm-up call + timed call of tracking_leg), both of 200 iterations
d["_meta"] = {"track_iterations_per_launch": 200, "command": "python scripts/track_leg.py 200", "note": "FETCH_SIZE / WRITE_SIZE in KB as reported"}
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
PY
# the whole odometry loop
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_run.log 2>&1
cp $(find /tmp/p_odo -name "*kernel_stats.csv" | head -1) $OUT/odometry_kernel_stats.csv
# GPU timeline of the loop after the initialisation: busy time, gaps by length, kernel time per frame (same trace)
python scripts/odometry_timeline.py /tmp/p_odo $OUT/odometry_timeline.txt > /dev/null 2>&1
tail -1 $OUT/odometry_run.log > $OUT/odometry_loop_profiled.json
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop.json 2> $OUT/odometry_loop.err
# config 4's window at full size, both dtypes
for DT in f64 f32; do
  timeout 300 python bench.py --keyframes 32 --dtype $DT --no-cpu --no-secondary --steps 10 --warmup 2 > $OUT/bench_kf32_$DT.json 2> $OUT/bench_kf32_$DT.err
done
# config 5 on the one GPU (two ranks share it; gloo for the bracketing barrier)
COMO_SINGLE_DEVICE=1 COMO_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --replicas --gpus 2 --steps 60 --warmup 5 > $OUT/bench_replicas2_one_gpu.json 2> $OUT/bench_replicas2_one_gpu.err
timeout 300 python bench.py --replicas --steps 60 --warmup 5 > $OUT/bench_replicas1.json 2> $OUT/bench_replicas1.err
# round 4: solver / conditioning / network timings (graph replay, HIP events) and the 300-frame loop
for L in 1 0; do echo "COMO_CHOL_LEAN=$L"; COMO_CHOL_LEAN=$L timeout 300 python scripts/chol_time.py 200 760 1240 2680; done > $OUT/chol_time.txt 2>&1
for F in 1 0; do echo "COMO_CHOL_SMALL_FAST=$F"; COMO_CHOL_SMALL_FAST=$F timeout 120 python scripts/chol_small_time.py; done > $OUT/chol_small_time.txt 2>&1
timeout 300 python scripts/nn_time.py --layers > $OUT/nn_time.txt 2>&1
COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 > $OUT/odometry_loop300.json 2>> $OUT/odometry_loop.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_chol -- python scripts/chol_time.py 760 > $OUT/chol_stats_run.log 2>&1
cp $(find /tmp/p_chol -name "*kernel_stats.csv" | head -1) $OUT/chol_kernel_stats.csv
# the default bench line on the same box
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
head -8 $OUT/bench_f64_eager_kernel_stats.csv | cut -c1-140
grep -i "pair2\|dense_ref\|residual\|track_level" $OUT/bench_pmc_summary.txt $OUT/track_pmc_summary.txt | head
grep -i "pair2" $OUT/bench_mfma_summary.txt | head -3
cut -c1-400 $OUT/odometry_loop.json; echo
for f in $OUT/bench_kf32_f64.json $OUT/bench_kf32_f32.json $OUT/bench_replicas2_one_gpu.json $OUT/bench_replicas1.json; do tail -1 $f | cut -c1-330; echo; done
tail -1 $OUT/bench_line.json | cut -c1-600
grep -v amdgpu $OUT/chol_time.txt $OUT/chol_small_time.txt | head -30; head -4 $OUT/nn_time.txt; cut -c1-200 $OUT/odometry_loop300.json
