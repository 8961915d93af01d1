cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4d4
for P in 0 1; do echo "COMO_SIDE_PRIORITY=$P"; COMO_SIDE_PRIORITY=$P timeout 300 python bench.py --no-cpu --no-secondary 2>/dev/null | tail -1 | cut -c1-330; echo; done | tee gpurun_out/r4d4/bench_ab.txt
