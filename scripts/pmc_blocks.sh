#!/bin/bash
# GPU box: instruction-mix / stall counters of the two block kernels (one --pmc group per pass).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_blocks; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
wc -l $OUT/sq_counters.txt
for DT in f32 f64; do
  CMD="python bench.py --steps 6 --warmup 2 --no-cpu --eager --no-secondary --dtype $DT"
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pb_${DT}_$i -- $CMD > $OUT/run_${DT}_$i.log 2>&1
  done
done
mkdir -p /tmp/pb_all; k=0; for f in $(find /tmp/pb_f32_* /tmp/pb_f64_* -name "*counter_collection.csv"); do k=$((k+1)); cp $f /tmp/pb_all/${k}_counter_collection.csv; done
python scripts/pmc_summary.py /tmp/pb_all $OUT/summary.json 2>/dev/null | grep "pair2"
