#!/bin/bash
# Round 5 profile collection on the GPU box (through gpurun) -> gpurun_out/profiles_r5/ (the small summaries are copied into
# profiles/ as r5_* afterwards).  Counters in their own passes (--kernel-trace + --pmc only).
#   bash scripts/collect_profiles_r5.sh [bench] [aux] [odo] [line]        (default: all)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r5
mkdir -p $OUT
WHAT="${*:-bench aux odo line}"
has() { case " $WHAT " in *" $1 "*) return 0;; *) return 1;; esac; }
pmc_merge() {   # <out prefix> <dirs...>: counter files of several passes -> one summary
  local out=$1; shift; local d=/tmp/pm_$$_$RANDOM; mkdir -p $d; local i=0
  for f in $(find "$@" -name "*counter_collection.csv"); do i=$((i+1)); cp $f $d/${i}_counter_collection.csv; done
  python scripts/pmc_summary.py $d $out.json > $out.txt 2>&1
}
if has bench; then
  for DT in f64 f32; do
    CMD="python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype $DT"
    rm -rf /tmp/p_stats_$DT; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats_$DT -- $CMD > $OUT/bench_${DT}_stats_run.log 2>&1
    cp $(find /tmp/p_stats_$DT -name "*kernel_stats.csv" | head -1) $OUT/bench_${DT}_eager_kernel_stats.csv
  done
  CMD="python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype f64"
  rm -rf /tmp/p_f /tmp/p_w /tmp/p_m
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -- $CMD > $OUT/bench_f64_fetch_run.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -- $CMD > $OUT/bench_f64_write_run.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_m -- $CMD > $OUT/bench_f64_mfma_run.log 2>&1
  pmc_merge $OUT/bench_pmc_summary /tmp/p_f /tmp/p_w
  pmc_merge $OUT/bench_mfma_summary /tmp/p_m
  rm -rf /tmp/p_w4; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_w4 -- python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype f64 --window 4 > $OUT/bench_w4_stats_run.log 2>&1
  cp $(find /tmp/p_w4 -name "*kernel_stats.csv" | head -1) $OUT/bench_w4_f64_eager_kernel_stats.csv
fi
if has aux; then
  # dense solve: graph replay timings (persistent / multi-launch), in-kernel stamps of the chain workgroup, the hand-off protocols
  { echo "== default (persistent one-launch solver up to 34 column pairs, D <= 2175; the multi-launch solver beyond)"; timeout 200 python scripts/chol_time.py 200 760 1000 1240 1300 1500 2000 2680;
    echo "== COMO_CHOLP_MAX_NP=42 (persistent at D = 2680 too: slower than the multi-launch solver there)"; COMO_CHOLP_MAX_NP=42 timeout 200 python scripts/chol_time.py 2680;
    echo "== COMO_CHOLP=0 (multi-launch solver)"; COMO_CHOLP=0 timeout 200 python scripts/chol_time.py 200 760 1000 1240 1300 1500 2000; } > $OUT/chol_time.txt 2>&1
  for D in 760 1000 200 1300; do timeout 60 scripts/micro/bin/cholp_stamps $D; done > $OUT/cholp_stamps.txt 2>&1
  timeout 120 scripts/micro/bin/handoff > $OUT/handoff.txt 2>&1
  rm -rf /tmp/p_chol; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_chol -- python scripts/chol_time.py 760 > $OUT/chol_stats_run.log 2>&1
  cp $(find /tmp/p_chol -name "*kernel_stats.csv" | head -1) $OUT/chol_kernel_stats.csv
  # DepthCov network: wall time, per-kernel averages, matrix-pipe counters
  timeout 200 python scripts/nn_time.py --layers > $OUT/nn_time.txt 2>&1
  rm -rf /tmp/p_nn /tmp/p_nns; timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_nn -- python scripts/nn_time.py > $OUT/nn_mfma_run.log 2>&1
  pmc_merge $OUT/nn_mfma_summary /tmp/p_nn
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_nns -- python scripts/nn_time.py > $OUT/nn_stats_run.log 2>&1
  cp $(find /tmp/p_nns -name "*kernel_stats.csv" | head -1) $OUT/nn_kernel_stats.csv
  # tracking level kernel: one launch = 200 iterations, all four pyramid levels through bench's helper
  rm -rf /tmp/p_trk /tmp/p_trk_f /tmp/p_trk_w
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trk -- python scripts/track_leg.py 200 > $OUT/track_stats_run.log 2>&1
  cp $(find /tmp/p_trk -name "*kernel_stats.csv" | head -1) $OUT/track_kernel_stats.csv
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_trk_f -- python scripts/track_leg.py 200 > $OUT/track_fetch_run.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_trk_w -- python scripts/track_leg.py 200 > $OUT/track_write_run.log 2>&1
  pmc_merge $OUT/track_pmc_summary /tmp/p_trk_f /tmp/p_trk_w
  python - <<'PY'
import json
p = "gpurun_out/profiles_r5/track_pmc_summary.json"
d = json.load(open(p))
d["_meta"] = {"track_iterations_per_launch": 200, "command": "python scripts/track_leg.py 200", "note": "FETCH_SIZE / WRITE_SIZE in KB as reported"}
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
PY
  # config 4's window at full size; config 5 (one sequence per rank) at N = 1 and two ranks sharing the one GPU
  for DT in f64 f32; do
    timeout 300 python bench.py --keyframes 32 --dtype $DT --no-cpu --no-secondary --steps 10 --warmup 2 > $OUT/bench_kf32_$DT.json 2> $OUT/bench_kf32_$DT.err
  done
  COMO_SINGLE_DEVICE=1 COMO_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --replicas --gpus 2 --steps 60 --warmup 5 > $OUT/bench_replicas2_one_gpu.json 2> $OUT/bench_replicas2_one_gpu.err
  timeout 300 python bench.py --replicas --steps 60 --warmup 5 > $OUT/bench_replicas1.json 2> $OUT/bench_replicas1.err
fi
if has odo; then
  COMO_ODO_BREAKDOWN=1 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop_parts.json 2> $OUT/odo_parts.err
  rm -rf /tmp/p_odo; COMO_ODO_BREAKDOWN=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop_profiled.json 2> $OUT/odo.err
  python scripts/odometry_timeline.py /tmp/p_odo $OUT/odometry_timeline.txt > /dev/null 2>&1
  F=$(find /tmp/p_odo -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/odometry_kernel_stats.csv
  COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop.json 2>> $OUT/odo.err
  timeout 300 python scripts/kf_insert_profile.py --out $OUT/kf_insert.txt > /dev/null 2>> $OUT/odo.err
  COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 300 > $OUT/odometry_loop300.json 2>> $OUT/odo.err
fi
if has line; then
  timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
fi
head -8 $OUT/bench_f64_eager_kernel_stats.csv 2>/dev/null | cut -c1-140
grep -i "pair2\|dense_ref" $OUT/bench_pmc_summary.txt 2>/dev/null | head -4
grep -v amdgpu $OUT/chol_time.txt 2>/dev/null | head -14
grep "graph\|eager" $OUT/nn_time.txt 2>/dev/null
head -6 $OUT/odometry_timeline.txt 2>/dev/null
for f in $OUT/odometry_loop.json $OUT/odometry_loop300.json $OUT/bench_kf32_f64.json $OUT/bench_kf32_f32.json $OUT/bench_replicas2_one_gpu.json $OUT/bench_replicas1.json; do [ -f $f ] && { tail -1 $f | cut -c1-300; echo; }; done
[ -f $OUT/bench_line.json ] && tail -1 $OUT/bench_line.json | cut -c1-900
