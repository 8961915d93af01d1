#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for ch in 64 96 128 160; do
  COMO_BA_CHUNKS=$ch python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f32 chunks=$ch', round(d['value'],1), 'it/s  blocks', round(d['roofline']['kernel_ms']*1e3,1), 'us frac', round(d['roofline']['frac'],3))"
done
