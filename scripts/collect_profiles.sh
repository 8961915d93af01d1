#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM counters of the bench command -> gpurun_out/profiles_r1/
# (copy the small summaries into profiles/ afterwards).  Counters are collected in their own passes (kernel-trace only).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r1
mkdir -p $OUT
CMD="python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $CMD > $OUT/bench_stats_run.log 2>&1
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_eager_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- $CMD > $OUT/bench_fetch_run.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- $CMD > $OUT/bench_write_run.log 2>&1
mkdir -p /tmp/p_all
i=0; for f in $(find /tmp/p_fetch /tmp/p_write -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_all/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_all $OUT/bench_pmc_summary.json > $OUT/bench_pmc_summary.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_mfma -- $CMD > $OUT/bench_mfma_run.log 2>&1
mkdir -p /tmp/p_m
i=0; for f in $(find /tmp/p_mfma -name "*counter_collection.csv"); do i=$((i+1)); cp $f /tmp/p_m/${i}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/p_m $OUT/bench_mfma_summary.json > $OUT/bench_mfma_summary.txt 2>&1
tail -3 $OUT/bench_stats_run.log | cut -c1-300
head -12 $OUT/bench_eager_kernel_stats.csv | cut -c1-140
cat $OUT/bench_pmc_summary.txt | head -30
cat $OUT/bench_mfma_summary.txt | head -30
