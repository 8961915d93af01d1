#!/bin/bash
# GPU box: sweep the float64 two-pair block kernel over K~ ring depth (compile time) and pixel chunks per pair group.
cd "$GRAFT_REPO_ROOT"
for pf in 4 2 8; do
  COMO_EXTRA_HIPCC_FLAGS="-DCOMO_F64_PF=$pf" python -m como_amd.build --force > /dev/null 2>&1 || echo "build failed pf=$pf"
  for ch in 64 128 192 256; do
    COMO_BA_CHUNKS=$ch python bench.py --dtype f64 --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pf=$pf chunks=$ch', round(d['value'],1), 'it/s  blocks', round(d['roofline']['kernel_ms']*1e3,1), 'us frac', round(d['roofline']['frac'],3))"
  done
done
