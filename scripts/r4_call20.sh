timeout 600 python -m pytest tests/test_gpu_hotpath.py tests/test_gpu_r2.py tests/test_gpu_r4.py -q -x -k "median or select or fullsize_metric or double_select" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --no-secondary --no-cpu --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'blk_ms', round(d['roofline']['kernel_ms'],4))
"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_sel -- python bench.py --steps 20 --warmup 3 --no-cpu --eager --no-secondary --dtype f64 > /dev/null 2>&1
grep -i "select\|residual" $(find /tmp/p_sel -name "*kernel_stats.csv" | head -1) | cut -c1-150
