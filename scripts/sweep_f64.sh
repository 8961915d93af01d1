#!/bin/bash
# GPU box: float64 two-pair block kernel -- software-pipelined one wave per SIMD (variant 0) vs two waves per SIMD without the
# pipeline (COMO_BA_VARIANT=4), K~ ring depth (compile time) and pixel chunks per keyframe group.
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --dtype f64 --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'it/s  blocks', round(d['roofline']['kernel_ms']*1e3,1), 'us frac', round(d['roofline']['frac'],3), 'pose_err', d['solution']['max_pose_abs_err_vs_gt_end'])"; }
for pf in 4 2; do
  COMO_EXTRA_HIPCC_FLAGS="-DCOMO_F64_PF=$pf" python -m como_amd.build --force > /dev/null 2>&1 || echo "build failed pf=$pf"
  run "pf=$pf variant0"
  for ch in 96 128 192; do COMO_BA_VARIANT=4 COMO_BA_CHUNKS=$ch run "pf=$pf variant4 chunks=$ch"; done
done
COMO_BA_VARIANT=4 python -m pytest tests/test_gpu_hotpath.py -m gpu -q -x -k "window_iterate or factored or fullsize_window_properties" 2>&1 | tail -3
