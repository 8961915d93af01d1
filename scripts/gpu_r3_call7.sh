#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c7
mkdir -p $OUT
timeout 300 python scripts/debug_shard_equal.py > $OUT/debug_shard.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_r3.py -x -q -k "tracker_glue" > $OUT/pytest_a.log 2>&1; echo "rc $?" >> $OUT/pytest_a.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" $OUT/debug_shard.txt | tail -20
tail -5 $OUT/pytest_a.log | cut -c1-300
