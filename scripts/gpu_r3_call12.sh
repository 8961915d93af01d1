#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3c15
timeout 600 python -m pytest tests/test_gpu_r2.py -m gpu -q -k "fullsize_metric_window" > gpurun_out/r3c15/pytest_full.log 2>&1; tail -8 gpurun_out/r3c15/pytest_full.log
python -c "import json; [print({k:v for k,v in r.items() if k in (\"name\",\"pix\",\"window\",\"H_photo_scaled\",\"H_full\",\"probe_rel\")}) for r in json.load(open(\"gpurun_out/gpu_report.json\"))]"
COMO_ODO_BREAKDOWN=1 timeout 400 python scripts/gpu_odometry_bench.py --frames 100 > gpurun_out/r3c15/odo_parts.json 2> gpurun_out/r3c15/odo.err
cut -c1-1500 gpurun_out/r3c15/odo_parts.json
