#!/bin/bash
# GPU box: smoke runs of the bench modes the driver / configs 4 and 5 use.
cd "$GRAFT_REPO_ROOT"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), d['unit'], 'ms', round(d['ms_per_step'],3), 'graph', d['config']['hip_graph'], 'D', d['config']['system_dim'], 'scaling', d['scaling'], 'blocks_us', round(d['roofline']['kernel_ms']*1e3,1), 'info', d['solution']['cholesky_info'], 'pose_err', d['solution']['max_pose_abs_err_vs_gt_end'])"; }
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu --no-secondary 2>/dev/null | show torchrun1
python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu --no-secondary --force-shard 2>gpurun_out/force_shard.err | show force_shard
python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu --no-secondary --force-shard --eager 2>/dev/null | show force_shard_eager
python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu --no-secondary --replicas 2>/dev/null | show replicas
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --keyframes 32 2>gpurun_out/kf32.err | show kf32_dense
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --keyframes 32 --window 4 2>/dev/null | show kf32_w4
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --keyframes 32 --dtype f64 2>/dev/null | show kf32_dense_f64
tail -3 gpurun_out/force_shard.err gpurun_out/kf32.err
