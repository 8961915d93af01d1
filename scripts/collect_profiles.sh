#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM / MFMA counters of the bench command (both per-pixel
# dtypes), of the per-frame pieces (tracking level kernel, DepthCov network) and of the odometry loop -> gpurun_out/profiles_r2/
# (copy the small summaries into profiles/ afterwards).  Counters are collected in their own passes (kernel-trace only).
set -u
R=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
for DT in f32 f64; do
  CMD="python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype $DT"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats_$DT -- $CMD > $OUT/bench_${DT}_stats_run.log 2>&1
  cp $(find /tmp/p_stats_$DT -name "*kernel_stats.csv" | head -1) $OUT/bench_${DT}_eager_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch_$DT -- $CMD > $OUT/bench_${DT}_fetch_run.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write_$DT -- $CMD > $OUT/bench_${DT}_write_run.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma_$DT -- $CMD > $OUT/bench_${DT}_mfma_run.log 2>&1
done
mkdir -p /tmp/p_all /tmp/p_m
i=0; for f in $(find /tmp/p_fetch_f32 /tmp/p_write_f32 /tmp/p_fetch_f64 /tmp/p_write_f64 -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_all/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_all $OUT/bench_pmc_summary.json > $OUT/bench_pmc_summary.txt 2>&1
i=0; for f in $(find /tmp/p_mfma_f32 /tmp/p_mfma_f64 -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_m/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_m $OUT/bench_mfma_summary.json > $OUT/bench_mfma_summary.txt 2>&1
# per-frame pieces: tracking level kernel, DepthCov network, sampler, predictor
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_aux -- python scripts/gpu_aux_bench.py > $OUT/aux_run.log 2>&1
cp $(find /tmp/p_aux -name "*kernel_stats.csv" | head -1) $OUT/aux_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_nn -- python scripts/nn_profile.py > $OUT/nn_mfma_run.log 2>&1
mkdir -p /tmp/p_nn2; i=0; for f in $(find /tmp/p_nn -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_nn2/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_nn2 $OUT/nn_mfma_summary.json > $OUT/nn_mfma_summary.txt 2>&1
# the whole odometry loop
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_odo -- python scripts/gpu_odometry_bench.py --frames 100 > $OUT/odometry_run.log 2>&1
cp $(find /tmp/p_odo -name "*kernel_stats.csv" | head -1) $OUT/odometry_kernel_stats.csv
tail -1 $OUT/odometry_run.log > $OUT/odometry_loop.json
# the default bench line on the same box
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -2 $OUT/bench_f32_stats_run.log | cut -c1-300
head -8 $OUT/bench_f32_eager_kernel_stats.csv | cut -c1-140
head -6 $OUT/bench_f64_eager_kernel_stats.csv | cut -c1-140
grep -i "pair2\|dense_ref\|residual" $OUT/bench_pmc_summary.txt | head
grep -i "pair2" $OUT/bench_mfma_summary.txt | head
head -12 $OUT/aux_kernel_stats.csv | cut -c1-140
tail -1 $OUT/aux_run.log | cut -c1-600
cut -c1-300 $OUT/odometry_loop.json
