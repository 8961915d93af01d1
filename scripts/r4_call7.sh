mkdir -p gpurun_out/r4c7
for v in 13 15; do echo "COMO_BA_VARIANT=$v"; COMO_BA_VARIANT=$v timeout 600 python bench.py --no-secondary --no-cpu --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'blk_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3), 'pose_err', d['solution']['max_pose_abs_err_vs_gt_end'])
"; done 2>&1 | tee gpurun_out/r4c7/variants.txt
