#!/bin/bash
# Round 3, GPU call 2: new tests (compact path, small-system conditioning, sharded poison flags, config-5 replicas), the polished
# float64 block kernel (8- vs 4-deep K~ ring), the default bench line (float64 headline), then the whole GPU suite.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3c2
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_r3.py -x -q > $OUT/pytest_r3.log 2>&1; echo "rc $?" >> $OUT/pytest_r3.log
for V in 0 9 4; do
  COMO_BA_VARIANT=$V timeout 150 python bench.py --dtype f64 --no-cpu --no-secondary --steps 20 --warmup 3 > $OUT/bench_f64_v$V.json 2> $OUT/bench_f64_v$V.err
done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -k "replica" > $OUT/pytest_replicas.log 2>&1; echo "rc $?" >> $OUT/pytest_replicas.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_dist.py::test_bench_replicas_one_sequence_per_rank --deselect tests/test_gpu_dist.py::test_replica_sequences_share_one_gpu > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r3c2/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), "it/s %.1f ms %.3f blk_ms %.4f frac %.3f poseerr %.6e info %s" % (
            d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"],
            d["solution"]["max_pose_abs_err_vs_gt_end"], d["solution"]["cholesky_info"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", repr(e)[:100], open(f.replace(".json", ".err")).read()[-600:])
PY
tail -4 $OUT/pytest_r3.log; tail -4 $OUT/pytest_replicas.log; tail -5 $OUT/pytest.log
cut -c1-1500 $OUT/bench_default.json
