#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3c13
rocprofv3 --kernel-trace --stats -d gpurun_out/r3c13/ride -o t -- python scripts/chol_time.py 760 > gpurun_out/r3c13/ride.log 2>&1
COMO_CHOL_RIDE_MAX=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r3c13/noride -o t -- python scripts/chol_time.py 760 > gpurun_out/r3c13/noride.log 2>&1
for d in ride noride; do f=$(find gpurun_out/r3c13/$d -name "*kernel_stats.csv" | head -1); echo $d; grep -E "chol|Name" $f | cut -c1-160; done
