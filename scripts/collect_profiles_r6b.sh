#!/bin/bash
# Round 6, second collection (the files the tracker / frame-graph / hand-over work of the round's second half changed) ->
# gpurun_out/profiles_r6b/ (copied into profiles/ as r6b_*).  Every command under `timeout`.
#   bash scripts/collect_profiles_r6b.sh [odo] [ab] [track] [line]        (default: all)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r6b
mkdir -p $OUT
WHAT="${*:-odo ab track line}"
has() { case " $WHAT " in *" $1 "*) return 0;; *) return 1;; esac; }
if has odo; then
  rm -rf /tmp/p_odo; COMO_ODO_BREAKDOWN=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_loop_profiled.json 2> $OUT/odo.err
  python scripts/odometry_timeline.py /tmp/p_odo $OUT/odometry_timeline.txt > /dev/null 2>&1
  F=$(find /tmp/p_odo -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/odometry_kernel_stats.csv
  timeout 300 python scripts/frame_host_timeline.py $OUT/frame_host_timeline.txt > /dev/null 2>> $OUT/odo.err
fi
if has ab; then
  # medians per frame kind over 4 steady passes, one process per setting (scripts/loop_ab.py); the first process on a fresh box runs
  # ~3 % slow (clocks / page cache): a throw-away run first, the default measured first AND last
  timeout 200 python scripts/loop_ab.py 2 > /dev/null 2>&1
  for SW in "" "COMO_MEDIAN_AHEAD_MODE=off" "COMO_TRACK_LEAN_HEAD=0" "COMO_KF_IMAGES_AHEAD=0" "COMO_MEDIAN_AHEAD_MODE=off COMO_TRACK_LEAN_HEAD=0 COMO_KF_IMAGES_AHEAD=0" ""; do
    echo "== switches: ${SW:-defaults}" >> $OUT/odometry_loop_ab.txt
    env $SW timeout 300 python scripts/loop_ab.py 5 2>> $OUT/odo.err | tail -1 >> $OUT/odometry_loop_ab.txt
  done
fi
if has track; then
  { echo "== default (exact form)"; timeout 100 python scripts/track_leg.py 2>/dev/null | grep -o "us_per_iter_by_level[^}]*}";
    echo "== COMO_TRACK_SPLIT=1 (band-split sums)"; COMO_TRACK_SPLIT=1 timeout 100 python scripts/track_leg.py 2>/dev/null | grep -o "us_per_iter_by_level[^}]*}";
    echo "== COMO_TRACK_LOCAL=0 (device-wide form at every level)"; COMO_TRACK_LOCAL=0 timeout 100 python scripts/track_leg.py 2>/dev/null | grep -o "us_per_iter_by_level[^}]*}"; } > $OUT/track_levels.txt
  if [ -f como_amd/lib_prof/libcomo_hip.so ]; then      # (-DCOMO_TL_PROFILE build of csrc/track.hip: in-kernel phase stamps)
    for s in "480 640" "240 320" "120 160"; do
      echo "== exact form"; COMO_HIP_LIB=$PWD/como_amd/lib_prof/libcomo_hip.so timeout 100 python scripts/track_stamps.py $s 2>&1 | grep -v amdgpu | tail -15
      echo "== band-split sums"; COMO_TRACK_SPLIT=1 COMO_HIP_LIB=$PWD/como_amd/lib_prof/libcomo_hip.so timeout 100 python scripts/track_stamps.py $s 2>&1 | grep -v amdgpu | tail -15
    done > $OUT/track_stamps.txt
  fi
fi
if has line; then
  timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
fi
head -6 $OUT/odometry_timeline.txt 2>/dev/null
cat $OUT/odometry_loop_ab.txt $OUT/track_levels.txt 2>/dev/null | grep -v amdgpu
[ -f $OUT/bench_line.json ] && tail -1 $OUT/bench_line.json | cut -c1-400
