#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3c23
timeout 600 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py -m gpu -q -k "cholesky" > gpurun_out/r3c23/pytest_chol.log 2>&1; tail -4 gpurun_out/r3c23/pytest_chol.log
timeout 300 python scripts/chol_time.py 760 200 1240 2680 2>&1 | grep "D=" | tee gpurun_out/r3c23/chol.log
timeout 300 python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 --window 4 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('w4 f64', round(d['value'],1), round(d['ms_per_step'],4))"
