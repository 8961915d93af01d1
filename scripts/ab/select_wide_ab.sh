#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py tests/test_gpu_r3.py -m gpu -q -k "select or median" > gpurun_out/ab/pytest.log 2>&1; tail -4 gpurun_out/ab/pytest.log
run() { timeout 300 python bench.py --no-cpu --no-secondary --steps 30 --warmup 5 $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],4))"; }
for rep in 1 2; do
COMO_SEL_WIDE=1 run wide_f64 "--dtype f64"
COMO_SEL_WIDE=0 run narrow_f64 "--dtype f64"
COMO_SEL_WIDE=1 run wide_w4_f64 "--dtype f64 --window 4"
COMO_SEL_WIDE=0 run narrow_w4_f64 "--dtype f64 --window 4"
COMO_SEL_WIDE=1 run wide_f32 "--dtype f32"
COMO_SEL_WIDE=0 run narrow_f32 "--dtype f32"
done 2>&1 | tee gpurun_out/ab/ab.log
