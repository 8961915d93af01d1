#!/bin/bash
# Round 6 profile collection on the GPU box (through gpurun) -> gpurun_out/profiles_r6/ (the small summaries are copied into
# profiles/ as r6_* afterwards).  Counters in their own passes (--kernel-trace + --pmc only); every command under `timeout`.
#   bash scripts/collect_profiles_r6.sh [bench] [odo] [aux] [line]        (default: all)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r6
mkdir -p $OUT
WHAT="${*:-bench odo aux line}"
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
if has odo; then
  # the loop: kernel trace -> timeline by frame kind (dispatch counts, GPU busy time), per-kernel stats, the unprofiled rate
  rm -rf /tmp/p_odo; COMO_ODO_BREAKDOWN=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop_profiled.json 2> $OUT/odo.err
  python scripts/odometry_timeline.py /tmp/p_odo $OUT/odometry_timeline.txt > /dev/null 2>&1
  F=$(find /tmp/p_odo -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/odometry_kernel_stats.csv
  timeout 300 python scripts/loop_passes.py > $OUT/odometry_loop_passes.txt 2>> $OUT/odo.err
  for SW in "" "COMO_BA_RETARGET=0 COMO_FUSED_FRAME=0 COMO_KF_KEPT_MEDIANS=0 COMO_KF_HOST_CORR=0 COMO_KF_MASKED_DENSE=0 COMO_KF_KT_DIRECT=0 COMO_GREEDY_PERSIST=0 COMO_GREEDY_THIN_WAVE=0 COMO_BA_ASM_GROUPED=0 COMO_BA_FUSE_PASS1=0 COMO_SE3_NORMALIZE_KERNEL=0 COMO_KF_ASYNC_NETWORK=0 COMO_BA_SPECULATE=0 COMO_TRACK_REF_PIX=0"; do
    echo "== switches: ${SW:-defaults}" >> $OUT/odometry_loop_ab.txt
    for i in 1 2; do env $SW COMO_ODO_BREAKDOWN=0 timeout 300 python scripts/gpu_odometry_bench.py --frames 100 2>> $OUT/odo.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['loop_fps_after_init'],1), {k:(round(v['median'],2)) for k,v in d['frame_ms_by_request'].items()}, d['slowest_frames_ms'][:2])" >> $OUT/odometry_loop_ab.txt; done
  done
  timeout 200 python scripts/sampler_time.py > $OUT/sampler_time.txt 2>> $OUT/odo.err
  timeout 300 python scripts/frame_host_timeline.py $OUT/frame_host_timeline.txt > /dev/null 2>> $OUT/odo.err
fi
if has aux; then
  { echo "== default"; timeout 200 python scripts/chol_time.py 200 760 1000 1300 2000 2680; } > $OUT/chol_time.txt 2>&1
  timeout 200 python scripts/nn_time.py --layers > $OUT/nn_time.txt 2>&1
  for DT in f64 f32; do
    timeout 300 python bench.py --keyframes 32 --dtype $DT --no-cpu --no-secondary --steps 10 --warmup 2 > $OUT/bench_kf32_$DT.json 2> $OUT/bench_kf32_$DT.err
  done
  timeout 300 python bench.py --replicas --steps 60 --warmup 5 > $OUT/bench_replicas1.json 2> $OUT/bench_replicas1.err
fi
if has line; then
  timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
fi
head -8 $OUT/bench_f64_eager_kernel_stats.csv 2>/dev/null | cut -c1-140
grep -i "pair2\|dense_ref" $OUT/bench_pmc_summary.txt 2>/dev/null | head -4
head -6 $OUT/odometry_timeline.txt 2>/dev/null
cat $OUT/odometry_loop_ab.txt $OUT/odometry_loop_passes.txt 2>/dev/null | grep -v amdgpu
[ -f $OUT/bench_line.json ] && tail -1 $OUT/bench_line.json | cut -c1-600
